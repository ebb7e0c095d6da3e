#!/usr/bin/env python3
"""Kernel exploration on the GPU box: sweep rate per kernel / tuning / size.
Writes one line per variant; interleaves variants over rounds (one process)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import effectivediffusivityfvm_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,4096")
    ap.add_argument("--sweeps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="explicit,matfree,"
                                          "matfree_tb:tb_T=2,matfree_tb:tb_T=4,matfree_tb:tb_T=6,matfree_tb:tb_T=8,"
                                          "matfree_tb:tb_T=4:tb_LY=32,matfree_tb:tb_T=4:tb_LY=64,"
                                          "matfree_tb:tb_T=8:tb_LY=64,matfree_tb:tb_T=8:tb_LY=128")
    args = ap.parse_args()
    for n in [int(v) for v in args.sizes.split(",")]:
        solvers = []
        for var in args.variants.split(","):
            parts = var.split(":")
            kvs = dict(kv.split("=") for kv in parts[1:])
            nimg = int(kvs.pop("nimg", 1))
            s = pkg.Solver(n, n, kernel=parts[0], nimg=nimg)
            for k, v in kvs.items():
                s.set_tuning(k, int(v))
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(10)
            solvers.append((var, s, [], nimg))
        for _ in range(args.rounds):
            for var, s, times, nimg in solvers:
                times.append(s.sweeps(args.sweeps) / args.sweeps)   # ms per sweep
        for var, s, times, nimg in solvers:
            best, med = min(times), sorted(times)[len(times) // 2]
            rate = nimg * n * n / (med * 1e-3) / 1e6
            pl = s.plan()
            print(f"n={n:6d} {var:40s} med {med*1e3:9.2f} us  min {best*1e3:9.2f} us  "
                  f"{rate:10.0f} Mcells*iter/s  frac@64B {rate*64/8e6:.3f}  "
                  f"T={pl['tb_T']} LY={pl['tb_LY']} strips={pl['tb_strips']} cpi={pl['tb_chunks_per_image']} impl={pl['tb_impl']} R={pl['tb_R']} NW={pl['tb_NW']} wg={pl['tb_blocks']}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
