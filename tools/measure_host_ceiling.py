#!/usr/bin/env python3
"""Host-side ceiling of dataset generation (GPU box): how many 1024^2 JPEGs per second can deff2d's workers decode,
flood-fill, upload and report when the solve itself is next to nothing (MaxIter: 1 = one sweep + one check per image)?
BASELINE config #5 on 8 GPUs needs 8 x the one-GPU solve rate (~126 images/s of 10 001 sweeps each) ~ 1 000
images/s from the host side of ONE process; here 1, 2, 4 and 8 workers share one GPU (--devices 0,0,...), so the GPU work per
image is the same as it would be on 8 GPUs and what is measured is the host pipeline.
  python tools/measure_host_ceiling.py [N images] [size]"""
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
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out = {"image_size": S, "images": N, "host_cpus": os.cpu_count()}
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    for k in range(N):
        f = np.kron(rng.random((S // 8, S // 8)), np.ones((8, 8)))           # 8x8-pixel grains
        pix = np.where(f < rng.uniform(0.45, 0.75), 0, 255).astype(np.uint8)
        Image.fromarray(pix).save(os.path.join(d, f"{k:05d}.jpg"), quality=95)
    out["jpeg_bytes_mean"] = sum(os.path.getsize(os.path.join(d, f"{k:05d}.jpg")) for k in range(N)) / N
    out["generation_seconds"] = time.perf_counter() - t0
    open(os.path.join(d, "input.txt"), "w").write(
        "Input File:\nPhases: 2\nDs: 1e-3\nDf: 1\nMeshAmpX: 1\nMeshAmpY: 1\nCR: 1\nCL: 0\nOutputName: out.csv\n"
        f"printCMap: 0\nConvergence: 1e-6\nMaxIter: 1\nVerbose: 0\nRunBatch: 1\nNumImages: {N}\n")
    for threads in (1, 2, 3):
        for workers in (1, 2, 4, 8):
            extra = ["--devices", ",".join(["0"] * workers), "--prefetch-threads", str(threads)]
            t0 = time.perf_counter()
            r = subprocess.run([EXE, "input.txt", "--json", "w.json"] + extra, cwd=d, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            res = json.load(open(os.path.join(d, "w.json")))["results"]
            assert len(res) == N and all(x["iterations"] == 1 for x in res)
            out[f"workers_{workers}_prefetch_threads_{threads}"] = {"seconds": round(dt, 3), "images_per_s": round(N / dt, 1)}
print(json.dumps(out, indent=1))
