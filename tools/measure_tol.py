#!/usr/bin/env python3
"""Iterations-to-tolerance of BASELINE configs #2/#3 with the reference's stopping rule
(tol 1e-6, checks every 10 000 sweeps) and a raised MaxIter (run on the GPU box)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402

out = {}
for n, cap in ((1024, 30_000_000), (4096, 8_000_000)):
    with pkg.Solver(n, n) as s:
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        hist = []
        s.set_progress(lambda it, d, ch: hist.append((it, d, ch)) if it % 1_000_000 == 0 else None)
        t0 = time.perf_counter()
        r = s.solve(1e-6, cap)
        out[str(n)] = {"iters": r.iters, "hit_cap": r.iters >= cap, "deff": r.deff_raw, "conv": r.conv,
                       "wall_s": time.perf_counter() - t0, "loop_ms": r.loop_ms,
                       "Mcells_iter_per_s": n * n * r.iters / (r.loop_ms * 1e-3) / 1e6,
                       "trace_every_1e6": hist[:40]}
        print(json.dumps({str(n): out[str(n)]}), flush=True)
