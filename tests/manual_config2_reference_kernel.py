#!/usr/bin/env python3
"""One-off evidence run (GPU box; not collected by pytest: ~2 minutes): BASELINE config #2 IN FULL through the reference's own
code.  1024^2 synthetic image -> the reference's DiscretizeMatrix2D (oracle/_ref/ref_host) -> 1 970 001 launches of the
reference's updateX_SOR on the MI355X (oracle/_ref/ref_kernel; the count at which the reference's stopping rule fires,
as deff_solve reproduces it) against deff_solve(tol 1e-6) on the resident tiles: same field (SHA-256), and the Deff evaluated
from the reference kernel's field by the oracle's restatement of cuh:1252-1263 equals the solver's.
    python tests/manual_config2_reference_kernel.py > profiles/r04_config2_reference_kernel.json"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import oracle_binding as ob  # noqa: E402
import effectivediffusivityfvm_amd as pkg  # noqa: E402

n = 1024
pix = ob.synth_mask(n, n, 12345, 0)
D = ob.fill_D_2phase(pix, 1.0, 1e-3)
with pkg.Solver(n, n) as s:
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    t0 = time.perf_counter()
    r = s.solve(1e-6, 30_000_000)
    t_solver = time.perf_counter() - t0
    got = s.get_field()
    plan = s.plan()
A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir="/tmp")
t0 = time.perf_counter()
want, loop_ms = ob.ref_sweeps(A, b, ob.linear_guess(n, n, 0.0, 1.0), int(r.iters), timing=True, tmpdir="/tmp")
t_ref = time.perf_counter() - t0
deff_ref = ob.flux_deff(want, D, 0.0, 1.0)[0]
out = {"config": "BASELINE #2: ONE 1024x1024 synthetic image (seed 12345), Ds 1e-3, Df 1, tol 1e-6, check every 10 000 sweeps",
       "solver": {"iters": int(r.iters), "deff": r.deff_raw, "conv": r.conv, "seconds": round(t_solver, 2), "plan": plan,
                  "sha256": hashlib.sha256(got.tobytes()).hexdigest()},
       "reference_kernel": {"launches": int(r.iters), "loop_seconds": round(loop_ms / 1e3, 2), "wall_seconds": round(t_ref, 2),
                            "deff_from_its_field": deff_ref, "sha256": hashlib.sha256(want.tobytes()).hexdigest()},
       "field_bit_identical": bool(np.array_equal(got, want)), "deff_bit_identical": bool(deff_ref == r.deff_raw)}
print(json.dumps(out, indent=1))
sys.exit(0 if out["field_bit_identical"] and out["deff_bit_identical"] else 1)
