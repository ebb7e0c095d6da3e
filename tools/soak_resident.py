#!/usr/bin/env python3
"""Soak of the resident exchange (GPU box): for each size, `--sweeps` sweeps from the linear guess with resident passes
(flag-synchronised halo exchange, thousands of passes per launch) and again with one launch per pass; the SHA-256 of the two
fields must agree.  One JSON line per size.   python tools/soak_resident.py [--sweeps N] [sizes...]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import effectivediffusivityfvm_amd as pkg  # noqa: E402

args = sys.argv[1:]
sweeps = 4_000_000
if "--sweeps" in args:
    i = args.index("--sweeps")
    sweeps = int(args[i + 1])
    del args[i:i + 2]
ok = True
for n in [int(a) for a in args] or [1024, 1536, 2048]:
    out = {"n": n, "sweeps": sweeps}
    for tag, launch in (("resident", 0), ("one_launch_per_pass", 1)):
        with pkg.Solver(n, n, kernel="matfree_tb") as s:
            s.set_tuning("tb_launch", launch)
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            t0 = time.perf_counter()
            done = 0
            while done < sweeps:                       # in pieces, so that a long run reports progress
                k = min(1_000_000, sweeps - done)
                s.sweeps(k)
                done += k
            dt = time.perf_counter() - t0
            p = s.plan()
            out[tag] = {"sha256": hashlib.sha256(s.get_field().tobytes()).hexdigest(), "seconds": round(dt, 2),
                        "G_cells_iter_per_s": round(n * n * sweeps / dt / 1e9, 1), "NW": p["tb_NW"], "R": p["tb_R"], "T": p["tb_T"],
                        "resident": p["tb_resident"], "fallbacks": s.plan_value("tb_fallbacks")}
    out["equal"] = out["resident"]["sha256"] == out["one_launch_per_pass"]["sha256"]
    ok = ok and out["equal"] and out["resident"]["fallbacks"] == 0
    print(json.dumps(out), flush=True)
sys.exit(0 if ok else 1)
