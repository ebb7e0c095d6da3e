#!/usr/bin/env python3
"""Distribution of wave-tile start/end times inside one temporally blocked pass (GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402
from effectivediffusivityfvm_amd import _capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with pkg.Solver(n, n) as s:
    if T:
        s.set_tuning("tb_T", T)
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(48)
    L = _capi.load()
    nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    st, en = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
    ok = en > 0
    t0 = st[ok].min()
    st, en = (st[ok] - t0) / 100.0, (en[ok] - t0) / 100.0          # microseconds
    dur = en - st
    print(f"n={n} tiles={ok.sum()} launch span {en.max():.1f} us")
    for name, v in (("start", st), ("end", en), ("duration", dur)):
        q = np.percentile(v, [0, 5, 25, 50, 75, 95, 100])
        print(f"  {name:9s} min {q[0]:7.1f}  p5 {q[1]:7.1f}  p25 {q[2]:7.1f}  p50 {q[3]:7.1f}  p75 {q[4]:7.1f}  p95 {q[5]:7.1f}  max {q[6]:7.1f}")
    print(f"  mean duration / span = {dur.mean() / en.max():.3f}")
