#!/usr/bin/env python3
"""Where a resident pass spends its time (GPU box): per tile the 100 MHz wall clock at entry and, for the first three
passes of a launch, at {neighbours' flags seen, halo in registers, T sweeps done, rim stored + flag raised}
(deff_debug_tb_stamps on a resident plan: 12 stamps per tile).
  python tools/wgr_stamps.py [n] [tb_NW] [tb_R]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402
from effectivediffusivityfvm_amd import _capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with pkg.Solver(n, n) as s:
    if nw:
        s.set_tuning("tb_impl", 2)
        s.set_tuning("tb_NW", nw)
    if R:
        s.set_tuning("tb_R", R)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(960)
    ms = min(s.sweeps(4800) for _ in range(3))
    p = s.plan()
    print(f"n={n} plan NW={p['tb_NW']} R={p['tb_R']} LY={p['tb_LY']} tiles={p['tb_strips']}x{p['tb_chunks_per_image']} resident={p['tb_resident']}: "
          f"{ms / (4800 / p['tb_T']) * 1e3:.2f} us per pass of {p['tb_T']} sweeps = {n * n * 4800 / ms / 1e6:.0f} G cells*iter/s")
    if not p["tb_resident"]:
        sys.exit(0)
    L = _capi.load()
    nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    tiles = p["tb_strips"] * p["tb_chunks_per_image"]
    st = buf[:tiles * 12].reshape(tiles, 12).astype(np.int64)
    ok = st[:, 0] > 0
    st = (st[ok] - st[ok, 0].min()) / 100.0                          # microseconds
    med = lambda v: float(np.median(v))
    print(f"  tiles stamped {ok.sum()}; entry spread {st[:, 0].max():.2f} us")
    print(f"  pass 0: rows in {med(st[:, 1] - st[:, 0]):.2f}  sweeps {med(st[:, 2] - st[:, 1]):.2f}  rim+flag {med(st[:, 3] - st[:, 2]):.2f}")
    for q in (1, 2):
        b = 4 * q
        line = (f"  pass {q}: wait for neighbours {med(st[:, b] - st[:, b - 1]):.2f}  halo read {med(st[:, b + 1] - st[:, b]):.2f}  "
                f"sweeps {med(st[:, b + 2] - st[:, b + 1]):.2f} (min {np.min(st[:, b + 2] - st[:, b + 1]):.2f} max {np.max(st[:, b + 2] - st[:, b + 1]):.2f})")
        if q == 1:
            line += f"  rim+flag {med(st[:, b + 3] - st[:, b + 2]):.2f}  whole pass {med(st[:, b + 3] - st[:, b - 1]):.2f}"
        print(line)
