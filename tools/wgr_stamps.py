#!/usr/bin/env python3
"""Phase times inside a resident launch (kernels_wgtile.hpp, k_sweep_wgres), GPU box:  wgr_stamps.py n
   per tile, passes 0..2: neighbours seen / rows in / swept / stored + released (us from the first tile's entry)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402
from effectivediffusivityfvm_amd import _capi  # noqa: E402

n = int(sys.argv[1])
with pkg.Solver(n, n, kernel="matfree_tb") as s:
    s.set_tuning("tb_impl", 2)
    s.set_tuning("tb_T", 8)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(48)
    L = _capi.load()
    nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    p = s.plan()
    assert p["tb_resident"] == 1, p
    tiles = p["tb_strips"] * p["tb_chunks_per_image"]
    a = buf[: tiles * 12].astype(np.int64).reshape(tiles, 12)
    a = a[a[:, 10] > 0]
    a = (a - a[:, 0].min()) / 100.0
    names = ["entry", "p0 rows+lookups", "p0 swept", "p0 released", "p1 neighbours", "p1 rows", "p1 swept", "p1 released",
             "p2 neighbours", "p2 rows", "p2 swept", "p2 (last: no release)"]
    print(f"n={n} tiles={len(a)} of {tiles}; plan {p}")
    for k, name in enumerate(names):
        if k == 11:
            break
        q = np.percentile(a[:, k], [0, 50, 100])
        d = np.percentile(a[:, k] - a[:, k - 1], [0, 50, 100]) if k else q
        print(f"  {name:22s} at min {q[0]:7.2f} med {q[1]:7.2f} max {q[2]:7.2f}   step min {d[0]:6.2f} med {d[1]:6.2f} max {d[2]:6.2f}")
