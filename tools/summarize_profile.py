#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into the
files committed under profiles/: the rocprofv3 --stats kernel table and the HBM
traffic of the sweep kernel from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE tallies 128-B read requests at 64 B, i.e. reports half the bytes of a
coalesced stream (checked on k_make_c0 / k_init_linear, whose byte counts are
known exactly: 128 MiB read shows as 65 549 KiB), so reads = 2 * FETCH_SIZE;
WRITE_SIZE is exact.
usage: summarize_profile.py <tag> <round> <bench-kernel-name> <size>
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd, kname, size = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    def newest(pattern):
        return max(glob.glob(pattern), key=os.path.getmtime)

    stats = newest(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    sweep = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    pmc = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest(os.path.join(src, f"pmc_{ctr}", "*", "*_counter_collection.csv"))
        per = {}
        for r in csv.DictReader(open(f)):
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        pmc[ctr] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in per.items()}
    name = sweep["Name"]
    fetch = pmc["FETCH_SIZE"][name]["mean_KiB"] * 1024 * 2       # gfx950: reads counted at half
    write = pmc["WRITE_SIZE"][name]["mean_KiB"] * 1024
    summary = {
        "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py "
                   f"(tools/profile_bench.sh {tag})",
        "dominant_kernel": name,
        "calls": int(sweep["Calls"]),
        "avg_duration_us": float(sweep["AverageNs"]) / 1e3,
        "share_of_gpu_time_pct": float(sweep["Percentage"]),
        "fetch_bytes_per_launch_corrected": fetch,
        "write_bytes_per_launch": write,
        "hbm_bytes_per_launch": fetch + write,
        "cells": size * size,
        "bytes_per_cell_measured": (fetch + write) / (size * size),
        "pmc_raw_KiB": pmc,
    }
    json.dump(summary, open(os.path.join(dst, f"{rnd}_{tag}_summary.json"), "w"), indent=1)
    tfile = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
    traffic[f"{kname}_{size}"] = {"hbm_bytes_per_launch": fetch + write,
                                  "source": f"profiles/{rnd}_{tag}_summary.json"}
    json.dump(traffic, open(tfile, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "pmc_raw_KiB"}, indent=1))


if __name__ == "__main__":
    main()
