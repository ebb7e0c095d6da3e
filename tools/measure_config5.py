#!/usr/bin/env python3
"""BASELINE config #5 on ONE GPU: 1 024 synthetic 1024^2 images (SURVEY 8d generator, img = 0..1023),
a fixed 10 001 sweeps each (one check at sweep 1, one at 10 001), in stacks of `--stack` images.
Reports images/s and aggregate Mcells*iter/s end to end (image generation, assembly, solve, Deff)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=1024)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--stack", type=int, default=64)
ap.add_argument("--sweeps", type=int, default=10001)
ap.add_argument("--fma", type=int, default=0)
args = ap.parse_args()
n, B = args.size, args.stack
deffs = []
with pkg.Solver(n, n, nimg=B) as s:
    s.set_tuning("fma", args.fma)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(48)                                   # warm-up (kernels loaded, plan made)
    t0 = time.perf_counter()
    loop_ms = 0.0
    for g in range(0, args.images, B):
        s.synth_image(12345, g)                    # images g .. g+B-1
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(0.0, args.sweeps)            # tol 0: never converges, stops at MAX_ITER
        loop_ms += res[0].loop_ms
        deffs += [r.deff_raw for r in res]
        assert all(r.iters == args.sweeps for r in res)
    dt = time.perf_counter() - t0
print(json.dumps({"images": args.images, "size": n, "stack": B, "sweeps_per_image": args.sweeps, "fma": args.fma,
                  "seconds": dt, "images_per_s": args.images / dt,
                  "Mcells_iter_per_s_end_to_end": args.images * n * n * args.sweeps / dt / 1e6,
                  "Mcells_iter_per_s_solve_loops": args.images * n * n * args.sweeps / (loop_ms * 1e-3) / 1e6,
                  "deff_first": deffs[0], "deff_last": deffs[-1]}))
