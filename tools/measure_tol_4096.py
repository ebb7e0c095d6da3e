#!/usr/bin/env python3
"""BASELINE config #3, iterations to tolerance: 4096^2 synthetic image, tol 1e-6, the reference's rule
(check every 10 000 sweeps), MaxIter raised until a time budget runs out.  Writes a trace line every
2 M sweeps (keeps the run visibly alive) and stops itself before `--seconds`."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=1000.0)
ap.add_argument("--chunk", type=int, default=2_000_000)
ap.add_argument("--fma", type=int, default=0)
args = ap.parse_args()
n = 4096
with pkg.Solver(n, n) as s:
    s.set_tuning("fma", args.fma)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    last = {}
    s.set_progress(lambda it, d, ch: last.update(it=it, deff=d, change=ch))
    t0 = time.perf_counter()
    total = 0
    # a solve of `chunk` sweeps stops at its MaxIter; the next one warm-starts from the field.  Chunks are
    # multiples of 10 000 + 1 sweeps apart?  No: each solve restarts the reference's loop (check after its
    # sweep 1, deffOld = 5), so the sweep counts of the chunks add up but the check positions shift by one
    # sweep per chunk -- irrelevant for "how many sweeps until the change per 10 000 sweeps is < 1e-6".
    while time.perf_counter() - t0 < args.seconds:
        r = s.solve(1e-6, args.chunk)
        total += r.iters
        print(json.dumps({"sweeps": total, "deff": r.deff_raw, "change_per_10000": r.conv, "wall_s": time.perf_counter() - t0,
                          "converged": r.iters < args.chunk}), flush=True)
        if r.iters < args.chunk:
            break
