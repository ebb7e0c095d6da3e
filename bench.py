#!/usr/bin/env python3
"""Headline benchmark: Mcells*iter/s of the Jacobi sweep at 4096^2 on MI355X.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = `--sweeps-per-step` weighted-Jacobi sweeps plus one Deff evaluation
(wall fluxes) of one synthetic two-phase image (SURVEY.md 8d generator) that is
already resident in HBM.  With N > 1 every rank solves its own image (image
index = rank): the path shards as whole images, there is no data-path
collective, scaling is weak.  Rank 0 prints ONE JSON line.

roofline: algorithmic bytes = 64 B per cell per sweep (A 5x8 + b 8 + x 8 read,
xNew 8 written: the operator behind the reference's seam) x
cells x sweeps per launch, divided by the sweep kernel's average launch duration
measured with HIP events on the solver's own stream inside deff_sweeps().
cpu_baseline: the single-thread CPU oracle on a bounded sample of the same
workload (rank 0, N = 1 only); a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_CELL_SWEEP = 64.0


def cpu_baseline(n, seconds_target=12.0):
    """Single-thread oracle sweep rate on the same synthetic workload (bounded sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    ob.build()
    pix = ob.synth_mask(n, n, 12345, 0)
    D = ob.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.discretize(D, 0.0, 1.0)
    x = ob.linear_guess(n, n, 0.0, 1.0)
    t0 = time.perf_counter()
    x = ob.sweeps(A, b, x, 2)
    per = (time.perf_counter() - t0) / 2
    k = max(2, min(2000, int(seconds_target / max(per, 1e-9))))
    t0 = time.perf_counter()
    ob.sweeps(A, b, x, k)
    dt = time.perf_counter() - t0
    return {"value": n * n * k / dt / 1e6, "unit": "Mcells*iter/s", "cores": 1, "kind": "port",
            "sample": f"{k} sweeps of the {n}x{n} synthetic image, oracle/deff_oracle.c (gcc -O2, AoS, 1 thread), "
                      f"host has {os.cpu_count()} logical CPUs"}


def bench_slab(args, pkg, torch, dist, world, rank, local_rank, barrier, use_dist):
    """One --size x --size image split into `world` row slabs (RCCL halo exchange per blocked pass)."""
    n, S = args.size, args.sweeps_per_step
    uid = [pkg.rccl_unique_id() if rank == 0 else None]
    if use_dist:
        dist.broadcast_object_list(uid, src=0)
    s = pkg.SlabRank(n, n, rank, world, uid[0], device=local_rank)
    for kv in args.tune:
        k, v = kv.split("=")
        s.set_tuning(k, int(v))
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    for _ in range(args.warmup):
        s.sweeps(S, args.omega)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.sweeps(S, args.omega)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        cells = float(n) * n
        print(json.dumps({
            "metric": f"Mcells*iter/s (Jacobi sweep) at {n}^2, row slabs", "value": cells * S * args.steps / elapsed / 1e6,
            "unit": "Mcells*iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ONE {n}x{n} synthetic two-phase image split into {world} row slabs, 8-row halos, "
                                   f"RCCL send/recv once per temporally blocked pass; step = {S} sweeps",
                       "kernel": "matfree_tb", "sweeps_per_step": S}}), flush=True)
    s.close()
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--sweeps-per-step", type=int, default=1200)   # multiple of every T
    ap.add_argument("--kernel", default="auto", choices=["auto", "explicit", "scalar", "matfree", "matfree_tb"])
    ap.add_argument("--omega", type=float, default=2.0 / 3.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--explicit-sweeps", type=int, default=300,
                    help="sweeps of the secondary explicit-coefficient measurement (0 = skip)")
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning knob (repeatable)")
    ap.add_argument("--mode", default="images", choices=["images", "slab"],
                    help="images: one image per GPU, no collective (default, weak scaling); slab: ONE image of "
                         "--size rows split into row slabs over the GPUs with RCCL halo exchange (config #4, strong)")
    ap.add_argument("--batch", type=int, default=1, help="images per GPU swept together (dataset-generation mode)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch  # first: one HIP runtime for torch and libdeff_amd
    import torch.distributed as dist
    import effectivediffusivityfvm_amd as pkg

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (any N, also N = 1) use the process group: RCCL over xGMI
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    n, S = args.size, args.sweeps_per_step
    if args.mode == "slab":
        return bench_slab(args, pkg, torch, dist, world, rank, local_rank, barrier, use_dist)
    s = pkg.Solver(n, n, device=local_rank, kernel=args.kernel, nimg=args.batch)
    for kv in args.tune:
        k, v = kv.split("=")
        s.set_tuning(k, int(v))
    s.synth_image(12345, rank * args.batch)     # image index = rank (x batch): independent images, no comm
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)

    def step():
        ms = s.sweeps(S, args.omega)
        deff, _, _ = s.flux()
        return ms, (deff if args.batch == 1 else float(deff[0]))

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    deff = float("nan")
    for _ in range(args.steps):
        ms, deff = step()
        kernel_ms += ms
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    launches, sweeps_per_launch = s.last_launches()     # of the last step
    kernel_used = s.kernel_in_use()

    # Secondary, untimed-by-the-driver leg: the explicit-coefficient kernel, i.e. the
    # operator exactly as it sits behind the reference's seam (A, b streamed: 64 B/cell/sweep).
    explicit = None
    if args.explicit_sweeps > 0 and kernel_used != "explicit":
        s.set_kernel("explicit")
        s.sweeps(20, args.omega)
        ems = s.sweeps(args.explicit_sweeps, args.omega)
        explicit = (ems * 1e-3 / args.explicit_sweeps, s.kernel_in_use())
        s.set_kernel(args.kernel)

    # second reported row: plain Jacobi, omega = 1 (updateX_V1's arithmetic), same kernel
    omega1_ms = None
    if abs(args.omega - 1.0) > 1e-12:
        s.sweeps(sweeps_per_launch * 4, 1.0)
        omega1_ms = s.sweeps(S, 1.0)

    # third reported row: the same sweeps in contracted arithmetic (opt-in "fma" mode: products fused
    # into adds as a compiler contracts the reference's expression; bit-identical to the oracle's fma
    # build, NOT to the default arithmetic -- reported beside `value`, never as `value`)
    fma_ms = None
    if kernel_used != "explicit":
        s.set_tuning("fma", 1)
        s.sweeps(sweeps_per_launch * 4, args.omega)
        fma_ms = s.sweeps(S, args.omega)
        s.set_tuning("fma", 0)

    if rank == 0:
        cells = float(n) * n * args.batch
        total_launches = args.steps * launches
        launch_s = kernel_ms * 1e-3 / total_launches      # avg duration of one sweep-kernel launch (HIP events)
        alg_bytes = BYTES_PER_CELL_SWEEP * cells * sweeps_per_launch
        achieved = alg_bytes / launch_s / 1e9

        def traffic_of(kname):
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                return json.load(open(tfile)).get(f"{kname}_{n}")
            except Exception:
                return None

        out = {
            "metric": "Mcells*iter/s (Jacobi sweep) at 4096^2" if n == 4096 else f"Mcells*iter/s (Jacobi sweep) at {n}^2",
            "value": world * cells * S * args.steps / elapsed / 1e6,
            "unit": "Mcells*iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n}x{n} synthetic two-phase image (splitmix64 seed 12345, porosity 0.5), Ds=1e-3 Df=1 "
                            f"CL=0 CR=1, omega={args.omega:.6g}; step = {S} sweeps + 1 Deff evaluation; "
                            f"{args.batch} image(s) per GPU (image index = rank), no inter-GPU communication",
                "kernel": kernel_used,
                "sweeps_per_step": S,
                "sweeps_per_launch": sweeps_per_launch,
                "deff_raw_after_run": deff,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": kernel_used,
                "launch_us": launch_s * 1e6,
                "algorithmic_bytes_per_launch": alg_bytes,
                "model": "64 B per cell per sweep (A 40 + b 8 + x 8 read, xNew 8 written: the explicit operator "
                         "behind the reference seam, SURVEY.md 8d) x cells x sweeps per launch",
            },
        }
        if kernel_used in ("matfree", "matfree_tb"):
            # this kernel never materialises A: its own compulsory traffic is x 8 + code 2 read, xNew 8
            # written per cell per LAUNCH (a temporally blocked launch does sweeps_per_launch sweeps on it)
            own = 18.0 * cells
            if kernel_used == "matfree_tb":
                out["roofline"]["limiter"] = ("not HBM: FP64 issue ~50-55 % busy (an FP64 instruction issues once per "
                                              "8 clocks from one wave, the 4-clock rate needs two ready waves per SIMD; "
                                              "3-4 fit the registers), LDS row lookups ~40 %; profiles/r01d_tb_sq_counters.json, "
                                              "tools/ubench")
            out["roofline"]["own_model"] = {
                "bytes_per_launch": own,
                "achieved": own / launch_s / 1e9,
                "frac": own / launch_s / 1e9 / HBM_PEAK_GBS,
                "note": "matrix-free: coefficients come from a 16-bit row code through an LDS row dictionary; frac of the "
                        "64-B model above can exceed 1 because those bytes are never moved",
            }
        if omega1_ms:
            out["jacobi_omega_1"] = {"value": cells * S / (omega1_ms * 1e-3) / 1e6, "unit": "Mcells*iter/s",
                                     "sample": f"{S} sweeps with omega = 1 (plain Jacobi, updateX_V1), same image, one GPU"}
        if fma_ms:
            out["contracted_arithmetic"] = {
                "value": cells * S / (fma_ms * 1e-3) / 1e6, "unit": "Mcells*iter/s",
                "sample": f"{S} sweeps with deff_set_tuning('fma', 1): sigma += a*x as fma, final sum as fma "
                          "(7 instead of 11 FP64 instructions per cell); parity: bit-identical to the oracle built "
                          "with -ffp-contract=fast, reproduces the survey's primary recorded Deff for config #1"}
        tr = traffic_of(kernel_used)
        if tr:
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr.get("source")
        if explicit:
            el, ek = explicit
            ea = BYTES_PER_CELL_SWEEP * cells / el / 1e9
            out["explicit_operator"] = {
                "kernel": ek,
                "value": cells / el / 1e6,
                "unit": "Mcells*iter/s",
                "sample": f"{args.explicit_sweeps} sweeps, same image, same run",
                "roofline": {"bound": "hbm", "achieved": ea, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ea / HBM_PEAK_GBS, "launch_us": el * 1e6,
                             "algorithmic_bytes_per_launch": BYTES_PER_CELL_SWEEP * cells, "traffic": None},
            }
            tr = traffic_of(ek)
            if tr:
                out["explicit_operator"]["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["explicit_operator"]["roofline"]["traffic_source"] = tr.get("source")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out), flush=True)
    s.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
