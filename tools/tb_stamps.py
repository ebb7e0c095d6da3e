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
    for k, v in zip(("tb_rank_w0", "tb_rank_w1", "tb_rank_w2", "tb_rank_wall"), sys.argv[3:7]):
        s.set_tuning(k, int(v))
    if os.environ.get("TB_RANKED") is not None:
        s.set_tuning("tb_ranked", int(os.environ["TB_RANKED"]))
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    s.sweeps(48)
    L = _capi.load()
    nt = C.c_int()
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, None, C.byref(nt)))
    buf = np.zeros(2 * nt.value, dtype=np.uint64)
    _capi.check(L.deff_debug_tb_stamps(s._ctx, 2.0 / 3.0, buf.ctypes.data_as(C.c_void_p), C.byref(nt)))
    st = buf[0::2].astype(np.int64)
    dur_t = (buf[1::2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    where = (buf[1::2] >> np.uint64(32)).astype(np.int64)          # HW_ID[15:0] | XCC << 16
    ok = dur_t > 0
    t0 = st[ok].min()
    st, dur, where = (st[ok] - t0) / 100.0, dur_t[ok] / 100.0, where[ok]   # microseconds
    en = st + dur
    print(f"n={n} tiles={ok.sum()} launch span {en.max():.1f} us")
    for name, v in (("start", st), ("end", en), ("duration", dur)):
        q = np.percentile(v, [0, 5, 25, 50, 75, 95, 100])
        print(f"  {name:9s} min {q[0]:7.1f}  p5 {q[1]:7.1f}  p25 {q[2]:7.1f}  p50 {q[3]:7.1f}  p75 {q[4]:7.1f}  p95 {q[5]:7.1f}  max {q[6]:7.1f}")
    print(f"  mean duration / span = {dur.mean() / en.max():.3f}")
    # per SIMD (XCC, SE, CU, SIMD of HW_ID): when its waves end, oldest first
    simd = where >> 4                                              # drop the wave slot
    ids = np.unique(simd)
    ends = {k: np.sort(en[simd == k]) for k in ids}
    cnt = np.array([len(v) for v in ends.values()])
    last = np.array([v[-1] for v in ends.values()])
    print(f"  SIMDs seen {len(ids)}; waves per SIMD: " + ", ".join(f"{c} x{(cnt == c).sum()}" for c in np.unique(cnt)))
    for k in range(cnt.max()):
        v = np.array([e[k] for e in ends.values() if len(e) > k])
        print(f"  end of a SIMD's wave #{k + 1}: mean {v.mean():6.1f}  p5 {np.percentile(v, 5):6.1f}  p95 {np.percentile(v, 95):6.1f}")
    print(f"  a SIMD's last wave ends: mean {last.mean():.1f}  min {last.min():.1f}  max {last.max():.1f} us -> SIMD-time idle before the launch ends: {1 - last.mean() / en.max():.3f}")
    # does the wave slot (HW_ID[3:0]) tell the order in which a SIMD serves its waves?
    from collections import Counter
    order = Counter()
    for k in ids:
        m = simd == k
        order[tuple((where[m] & 15)[np.argsort(en[m])].tolist())] += 1
    print("  wave slots of a SIMD in the order its waves end: " + ", ".join(f"{k} x{v}" for k, v in order.most_common(8)))
    # ... and does the workgroup's index tell the slot?  (blockIdx = kk * 8 + xcd, workgroup tile bt = xcd * per + kk, 4 waves each)
    wt_all = np.nonzero(ok)[0]
    nbt = (nt.value + 3) // 4
    per = (nbt + 7) // 8
    kk = (wt_all // 4) % per
    slot = where & 15
    for sl in np.unique(slot):
        v = kk[slot == sl]
        print(f"  wave slot {sl}: workgroups kk (position in its XCD's dispatch order) min {v.min()} p5 {int(np.percentile(v, 5))} median {int(np.median(v))} p95 {int(np.percentile(v, 95))} max {v.max()}")
    # the stragglers: which tiles end last?
    plan = s.plan()
    gy = plan["tb_chunks_per_image"]
    late = np.argsort(en)[::-1][:24]
    if plan.get("tb_ranked"):                                       # dealt tiles: index = rank * strips * nq + tx * nq + q
        nq = plan["tb_chunks_per_image"] // 3
        per_cls = plan["tb_strips"] * nq
        print(f"  dealt tiles, {nq} chunks per rank and strip; last 24: " + "  ".join(f"[rank {wt_all[i] // per_cls} tx {(wt_all[i] % per_cls) // nq} q {wt_all[i] % nq} slot {where[i] & 15}: {en[i]:.1f}]" for i in late))
        for c in range(3):
            mm = wt_all // per_cls == c
            print(f"  rank {c}: {mm.sum()} tiles, end mean {en[mm].mean():.1f} p95 {np.percentile(en[mm], 95):.1f} max {en[mm].max():.1f}; slots " + str(sorted(set((where[mm] & 15).tolist()))))
        print("  the same with where they ran: " + "  ".join(f"[r{wt_all[i] // per_cls} tx{(wt_all[i] % per_cls) // nq} q{wt_all[i] % nq} xcc{where[i] >> 16} se{(where[i] >> 13) & 7} cu{(where[i] >> 8) & 15} simd{(where[i] >> 4) & 3}: {en[i]:.1f}]" for i in np.argsort(en)[::-1][:48]))
        cu_id = where >> 6                                         # XCC, SE, CU (drop SIMD and slot)
        cu_end = {k: en[cu_id == k].max() for k in np.unique(cu_id)}
        ce = np.array(sorted(cu_end.values()))
        print(f"  CUs {len(ce)}: last end of a CU p5 {np.percentile(ce, 5):.1f} median {np.median(ce):.1f} p95 {np.percentile(ce, 95):.1f} max {ce.max():.1f}; the 8 slowest: " + " ".join(f"{v:.1f}" for v in ce[-8:]))
        txd = (wt_all % per_cls) // nq
        print("  mean end by strip: " + " ".join(f"{t}:{en[txd == t].mean():.0f}" for t in np.unique(txd)))
        sys.exit(0)
    ntx_ = plan["tb_strips"]                                        # equal chunks, default numbering (tb_xmajor = 1): wt = chunk * strips + strip
    print("  last 24 tiles to end: " + "  ".join(f"[tx {wt_all[i] % ntx_} ch {wt_all[i] // ntx_} xcc {where[i] >> 16} cu {(where[i] >> 8) & 15} se {(where[i] >> 13) & 7} simd {(where[i] >> 4) & 3} slot {where[i] & 15}: {en[i]:.1f}]" for i in late))
    txs = wt_all % ntx_
    for name, m in (("first strip", txs == 0), ("last strip", txs == txs.max()), ("inner strips", (txs > 0) & (txs < txs.max()))):
        for sl in (0, 1, 2):
            mm = m & (slot == sl)
            if mm.any():
                print(f"  {name:12s} slot {sl}: {mm.sum():4d} tiles, duration mean {dur[mm].mean():6.1f} max {dur[mm].max():6.1f}")
    xcc = where >> 16
    for xc in np.unique(xcc):
        m = xcc == xc
        print(f"  XCC {xc}: tiles {m.sum():4d}  last end {en[m].max():6.1f}  mean end {en[m].mean():6.1f}")
