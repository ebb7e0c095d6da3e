#!/usr/bin/env python3
"""One-off measurements quoted in DESIGN.md (run on the GPU box):
host->device hand-over costs at the boundary, and iterations-to-tolerance of the
BASELINE configs #2 (1024^2) and #3 (4096^2) with the reference's stopping rule."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import effectivediffusivityfvm_amd as pkg  # noqa: E402

out = {}
for n in (1024, 4096):
    with pkg.Solver(n, n) as s:
        s.synth_image(12345, 0)
        pix = s.get_image()
        t0 = time.perf_counter(); s.set_image(pix); s.assemble_2phase(1e-3, 1.0, 0.0, 1.0); s.synchronize()
        t_native = time.perf_counter() - t0
        s.init_linear(0.0, 1.0)
        t0 = time.perf_counter()
        r = s.solve(1e-6, 500000)
        wall = time.perf_counter() - t0
        out[f"solve_{n}"] = {"iters": r.iters, "checks": r.checks, "deff": r.deff_raw / 1.0, "conv": r.conv,
                             "loop_ms": r.loop_ms, "wall_s": wall, "kernel": s.kernel_in_use(),
                             "Mcells_iter_per_s": n * n * r.iters / (r.loop_ms * 1e-3) / 1e6}
        out[f"handover_native_{n}"] = {"bytes": int(pix.size), "seconds_upload_plus_assembly": t_native}
        if n == 4096:
            s.set_kernel("explicit"); s.sweeps(0)
            A, b = s.get_system()
            D = np.where(pix < 150, 1.0, 1e-3)
            t0 = time.perf_counter(); s.set_system(A, b, D, 0.0, 1.0); s.synchronize()
            dt = time.perf_counter() - t0
            out["handover_set_system_4096"] = {"bytes": int(A.nbytes + b.nbytes), "seconds": dt,
                                               "GBps": (A.nbytes + b.nbytes) / dt / 1e9}
            x = np.empty((n, n)); t0 = time.perf_counter(); x = s.get_field(); dt = time.perf_counter() - t0
            out["field_download_4096"] = {"bytes": int(x.nbytes), "seconds": dt}
print(json.dumps(out, indent=1))
