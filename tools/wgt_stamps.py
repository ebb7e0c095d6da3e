#!/usr/bin/env python3
"""Phase times inside one workgroup-tile pass (kernels_wgtile.hpp), GPU box:
   wgt_stamps.py n T R NW  -> per tile: entry, dictionary loaded, rows loaded, after each sweep, stored (us)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402
from effectivediffusivityfvm_amd import _capi  # noqa: E402

n, T, R, NW = (int(v) for v in sys.argv[1:5])
with pkg.Solver(n, n, kernel="matfree_tb") as s:
    for k, v in (("tb_T", T), ("tb_impl", 2), ("tb_R", R), ("tb_NW", NW)):
        s.set_tuning(k, v)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(6 * T)
    L = _capi.load()
    nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    p = s.plan()
    tiles = p["tb_strips"] * p["tb_chunks_per_image"]
    a = buf[: tiles * (T + 4)].astype(np.int64).reshape(tiles, T + 4)
    ok = a[:, -1] > 0
    a = a[ok]
    t0 = a[:, 0].min()
    a = (a - t0) / 100.0
    names = ["entry", "lut", "rows"] + [f"sweep{t}" for t in range(1, T + 1)] + ["stored"]
    print(f"n={n} T={T} R={R} NW={NW} tiles={len(a)} of {tiles}; span {a[:, -1].max():.2f} us; plan {p}")
    for k, name in enumerate(names):
        q = np.percentile(a[:, k], [0, 50, 100])
        d = np.percentile(a[:, k] - a[:, k - 1], [0, 50, 100]) if k else q
        print(f"  {name:8s} at min {q[0]:7.2f} med {q[1]:7.2f} max {q[2]:7.2f}   step min {d[0]:6.2f} med {d[1]:6.2f} max {d[2]:6.2f}")
