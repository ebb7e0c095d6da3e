#!/usr/bin/env python3
"""Dataset-generation throughput of the command-line driver (GPU box): N synthetic two-phase
JPEGs of s x s pixels, the reference's batch mode (RunBatch: 1), tol 1e-6 / MaxIter 5e5."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "effectivediffusivityfvm_amd", "deff2d")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
out = {}
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    for k in range(N):
        f = rng.random((S // 8, S // 8))
        f = np.kron(f, np.ones((8, 8)))                      # 8x8-pixel grains
        pix = np.where(f < rng.uniform(0.45, 0.75), 0, 255).astype(np.uint8)
        Image.fromarray(pix).save(os.path.join(d, f"{k:05d}.jpg"), quality=95)
    open(os.path.join(d, "input.txt"), "w").write(
        "Input File:\nPhases: 2\nDs: 1e-3\nDf: 1\nMeshAmpX: 1\nMeshAmpY: 1\nCR: 1\nCL: 0\nOutputName: out.csv\n"
        f"printCMap: 0\nConvergence: 1e-6\nMaxIter: 5e5\nVerbose: 0\nRunBatch: 1\nNumImages: {N}\n")
    runs = [("grouped", [])]
    if N >= 12288:
        runs.append(("grouped_1024_slots", ["--batch-size", "1024"]))
    if os.environ.get("DEFF_TWO_WORKERS"):
        # two host threads + contexts + streams on ONE GPU: the tail of one stream's launch overlaps the other's head
        runs.append(("two_workers_one_gpu", ["--devices", "0,0"]))
        runs.append(("three_workers_one_gpu", ["--devices", "0,0,0"]))
    runs.append(("one_at_a_time", ["--batch-size", "1"]))
    for label, extra in runs:
        if label == "one_at_a_time" and N > 64:
            open(os.path.join(d, "input.txt"), "a").write("NumImages: 64\n")      # a later key wins: bounded sample
        t0 = time.perf_counter()
        r = subprocess.run([EXE, "input.txt", "--json", f"{label}.json"] + extra, cwd=d, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        res = json.load(open(os.path.join(d, f"{label}.json")))["results"]
        sweeps = sum(x["iterations"] for x in res)
        out[label] = {"images": len(res), "seconds": dt, "images_per_s": len(res) / dt, "total_sweeps": sweeps,
                      "Mcells_iter_per_s_end_to_end": sweeps * S * S / dt / 1e6,
                      "mean_iterations": sweeps / len(res)}
print(json.dumps({"image_size": S, **out}, indent=1))
