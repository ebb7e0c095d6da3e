import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effectivediffusivityfvm_amd as pkg
from effectivediffusivityfvm_amd import _capi
n = int(sys.argv[1])
with pkg.Solver(n, n) as s:
    s.synth_image(12345, 0); s.assemble_2phase(1e-3, 1.0, 0.0, 1.0); s.init_linear(0.0, 1.0)
    s.sweeps(64)
    p = s.plan(); print(n, p)
    L = _capi.load(); nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    tiles = p["tb_strips"] * p["tb_chunks_per_image"]
    st = buf[:tiles * 12].reshape(tiles, 12).astype(np.int64)
    ok = (st[:, 11] > 0) & (st[:, 11] < 10**7)
    st = st[ok]
    print("tiles", ok.sum(), " clocks from the previous barrier's release (as wave 0 saw it) to each wave's arrival at sweep 4's barrier, median over tiles:")
    for w in range(11):
        print(f"  wave {w:2d} (SIMD {w % 4}, rank {w // 4}): median {np.median(st[:, w]):7.0f}  p10 {np.percentile(st[:, w], 10):7.0f}  p90 {np.percentile(st[:, w], 90):7.0f}")
    print(f"  last of waves 11..15: median {np.median(st[:, 11]):7.0f}")
