#!/usr/bin/env python3
"""Large-image measurements on ONE GPU (quoted in DESIGN.md): 16384^2 as one context and as 4
row slabs on the same device (the slab path's own overhead: halo copies + smaller tiles)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402

out = {}
n, S = 16384, 240
with pkg.Solver(n, n) as s:
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(24)
    ms = min(s.sweeps(S) for _ in range(3))
    out["one_context_16384"] = {"us_per_sweep": ms * 1e3 / S, "Mcells_iter_per_s": n * n * S / (ms * 1e-3) / 1e6}
    r = s.solve(1e-6, 1)
    out["one_context_16384"]["first_check_deff"] = r.deff_raw
for k in (2, 4):
    with pkg.SlabGroup(n, n, [0] * k) as g:
        g.synth_image(12345, 0)
        g.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        g.init_linear(0.0, 1.0)
        g.sweeps(24)
        ms = min(g.sweeps(S) for _ in range(3))
        out[f"{k}_slabs_one_gpu_16384"] = {"us_per_sweep": ms * 1e3 / S,
                                           "Mcells_iter_per_s": n * n * S / (ms * 1e-3) / 1e6}
print(json.dumps(out, indent=1))
