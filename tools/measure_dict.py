#!/usr/bin/env python3
"""3-phase / host-assembled systems: explicit kernel vs harvested row dictionary (run on the GPU box)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402

out = {}
n = 4096
rng = np.random.default_rng(1)
# blobby three-level image: smooth noise thresholded into gas / fluid / solid
f = rng.random((n // 8, n // 8))
f = np.kron(f, np.ones((8, 8)))
pix = np.where(f < 0.33, 0, np.where(f < 0.66, 150, 255)).astype(np.uint8)
t0 = time.perf_counter()
grid, path = pkg.flood_fill((pix > 200).astype(np.uint32))
out["flood_fill_4096_s"] = time.perf_counter() - t0
for kernel in ("explicit", "auto"):
    with pkg.Solver(n, n, kernel=kernel) as s:
        s.set_image(pix)
        t0 = time.perf_counter()
        s.assemble_3phase(0.0, 1.0, 100.0, 0.0, 1.0, grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(12)
        setup = time.perf_counter() - t0
        ms = min(s.sweeps(240) for _ in range(3))
        out[f"3phase_{kernel}"] = {"kernel": s.kernel_in_use(), "us_per_sweep": ms * 1e3 / 240,
                                   "Mcells_iter_per_s": n * n * 240 / (ms * 1e-3) / 1e6, "setup_s": setup}
print(json.dumps(out, indent=1))
