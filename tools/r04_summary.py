#!/usr/bin/env python3
"""profiles/r04_* from gpurun_out/r04_evidence/ (what tools/r04_evidence.sh left): copies the logs and kernel stats and writes
profiles/r04_summary.json.   python tools/r04_summary.py"""
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "r04_evidence")
PROF = os.path.join(ROOT, "profiles")


def stats(path, needle):
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return {"avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "calls": int(r["Calls"])}
    return None


def json_line(path):
    for line in open(path, errors="replace"):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


b = json_line(os.path.join(EV, "bench_default.log"))
for src, dst in (("bench_default.log", "r04_bench_default.log"), ("tb_kernel_stats.csv", "r04_tb_kernel_stats.csv"),
                 ("explicit_kernel_stats.csv", "r04_explicit_kernel_stats.csv"),
                 ("residual_4096_kernel_stats.csv", "r04_residual_4096_kernel_stats.csv"),
                 ("residual_16384_kernel_stats.csv", "r04_residual_16384_kernel_stats.csv")):
    shutil.copyfile(os.path.join(EV, src), os.path.join(PROF, dst))
rf = b["roofline"]
res = json_line(os.path.join(EV, "residual_4096.log"))
traffic = {}
for line in open(os.path.join(EV, "residual_4096_traffic.txt")):
    m = re.match(r"(\w+) k_residual_classes mean ([\d.]+) KiB", line)
    if m:
        traffic[m.group(1)] = float(m.group(2))
hbm = (2 * traffic.get("FETCH_SIZE", 0.0) + traffic.get("WRITE_SIZE", 0.0)) * 1024
alg = 9 * 4096 * 4096
out = {
    "round": 4,
    "command": "tools/r04_evidence.sh on one MI355X (gpurun), then tools/r04_summary.py; final kernels of the round",
    "bench_value_Mcells_iter_per_s": b["value"], "ms_per_step": b["ms_per_step"], "plan": b["config"].get("plan"),
    "primary_kernel": {"name": rf.get("rocprof_kernel"), "events_launch_us": rf["launch_us"], "rocprof_avg_us": rf.get("rocprof_avg_us"),
                       "frac_fp64_valu": rf["frac"], "rocprof_frac": rf.get("rocprof_frac"), "traffic_bytes_per_launch": rf.get("traffic"),
                       "file": "profiles/r04_tb_kernel_stats.csv"},
    "explicit_kernel_64B": {"events_launch_us": rf.get("contract_64B_launch_us"), "rocprof_avg_us": rf.get("contract_64B_rocprof_avg_us"),
                            "frac_of_8TBs_events": rf.get("contract_64B_frac"), "frac_of_8TBs_rocprof": rf.get("contract_64B_rocprof_frac"),
                            "file": "profiles/r04_explicit_kernel_stats.csv"},
    "residual_4096": {"events_us_both_kernels": b["residual_4096"]["device_us"], "frac_of_8TBs": b["residual_4096"]["frac_of_hbm_peak"],
                      "under_rocprofv3_best_us": res.get("best_us"),
                      "rocprof_k_residual_classes": stats(os.path.join(EV, "residual_4096_kernel_stats.csv"), "k_residual_classes"),
                      "rocprof_k_residual_final": stats(os.path.join(EV, "residual_4096_kernel_stats.csv"), "k_residual_final"),
                      "pmc_FETCH_SIZE_KiB": traffic.get("FETCH_SIZE"), "pmc_WRITE_SIZE_KiB": traffic.get("WRITE_SIZE"),
                      "hbm_bytes_per_launch_corrected": hbm, "algorithmic_bytes": alg, "traffic_over_algorithmic": hbm / alg,
                      "note": "FETCH_SIZE x2 is the gfx950 correction of MI355X_MICROARCH.md; kernels run ~5-10 % slower under rocprofv3",
                      "file": "profiles/r04_residual_4096_kernel_stats.csv"},
    "residual_16384": {"rocprof_k_residual_classes": stats(os.path.join(EV, "residual_16384_kernel_stats.csv"), "k_residual_classes"),
                       "file": "profiles/r04_residual_16384_kernel_stats.csv"},
    "reference_kernel_on_this_gpu": b["cpu_baseline"].get("reference_kernel_on_this_gpu"),
    "cpu_baseline": {k: v for k, v in b["cpu_baseline"].items() if k != "reference_kernel_on_this_gpu"},
    "cpu_baseline_128": b["cpu_baseline_128"]["value"], "cpu_baseline_1024": b["cpu_baseline_1024"]["value"],
    "contracted_arithmetic": b.get("contracted_arithmetic"),
    "single_image_1024": b["single_image_1024"]["value"], "single_image_2048": b["single_image_2048"]["value"],
    "iters_to_tol_1024": b.get("iters_to_tol_1024"), "iters_to_tol_4096_cited": b.get("iters_to_tol_4096"),
}
with open(os.path.join(PROF, "r04_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(json.dumps({k: out[k] for k in ("bench_value_Mcells_iter_per_s", "ms_per_step", "plan")}))
print(json.dumps(out["primary_kernel"]))
