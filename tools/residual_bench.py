#!/usr/bin/env python3
"""Device time of deff_residual() (kernels_residual.hpp) by size: python tools/residual_bench.py [sizes...]
Prints one JSON line per size: microseconds (best / median of 20 calls, HIP events around the reduction's two kernels) and the
HBM rate against the algorithmic 9 B per cell (x 8 + pixel 1).  Under `rocprofv3 --kernel-trace --stats` the two kernels
(k_residual_classes, k_residual_final) show up separately."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import effectivediffusivityfvm_amd as pkg  # noqa: E402

args = sys.argv[1:]
kt = 0
if "--kt" in args:
    i = args.index("--kt")
    kt = int(args[i + 1])
    del args[i:i + 2]
for n in [int(a) for a in args] or [1024, 4096, 8192]:
    with pkg.Solver(n, n) as s:
        s.set_tuning("res_kt", kt)
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(8)
        for _ in range(3):
            s.residual()
        t = sorted(s.residual(timing=True)[1] * 1e3 for _ in range(20))
        best, med = t[0], statistics.median(t)
        print(json.dumps({"n": n, "kt": kt, "best_us": best, "median_us": med, "GBs_9B_per_cell": 9.0 * n * n / (best * 1e-6) / 1e9,
                          "frac_of_8TBs": 9.0 * n * n / (best * 1e-6) / 8e12, "residual": s.residual()}), flush=True)
