"""Dataset-generation (batch) mode across GPUs: whole images per rank, no data-path
collective (reference BatchSim, Deff2D.cuh:1843-2054).

Image k goes to rank k mod world; every rank reuses ONE solver context for all of
its images (the reference re-allocates and resets the device per image,
cuh:1975/cuh:2038); results land in the reference's NumImg x 9 table by image index,
so the output order does not depend on the number of ranks.  The only
communication is the final gather of that small table (torch.distributed: RCCL on
GPUs, gloo in the CPU tests).
"""
import numpy as np

# BatchSim's output row, cuh:2026-2034
COLUMNS = ("imgNum", "porosity", "PathFlag", "Deff", "Time", "nElements", "converge", "ds", "df")


def shard(num_images, rank, world):
    """Image indices owned by `rank` (round-robin, deterministic)."""
    return list(range(rank, num_images, world))


def solve_image(solver, pix, Ds, Df, CL, CR, tol, max_iter, omega=2.0 / 3.0):
    """One BatchSim iteration on an already-created context: image -> assembly ->
    linear guess -> solve.  Returns (deff normalised by Df (cuh:2017), conv, iters, loop_ms)."""
    solver.set_image(pix)
    solver.assemble_2phase(Ds, Df, CL, CR)
    solver.init_linear(CL, CR)                       # cuh:1955-1959
    r = solver.solve(tol, max_iter, omega=omega)
    return r.deff_raw / Df, r.conv, r.iters, r.loop_ms


def solve_image_3phase(solver, pix, Ds, Df, Dg, CL, CR, tol, max_iter, omega=2.0 / 3.0):
    """SingleSim3Phase / BatchSim3Phase (Deff2D.cuh:1316-1633, preCond = true cuh:1443): Grid from
    pixels > 200 + FloodFill; the gas diffusivity is ramped 10, 100, ... (< Dg) with the tolerance
    x10 and MAX_ITER 1e6 (the JacobiGPUPreCond stages, cuh:1486-1549), each stage warm-started from
    the previous field, which never leaves the device; then the real Dg with the user's tolerance
    (cuh:1553-1596).  Returns dict(deff, conv, stage_sweeps, path_flag, loop_ms)."""
    from .solver import flood_fill
    grid, path = flood_fill((np.asarray(pix) > 200).astype(np.uint32))
    solver.set_image(pix)
    solver.init_linear(CL, CR)
    stages, ms = [], 0.0
    g = 10.0
    while g < Dg:
        solver.assemble_3phase(Ds, Df, g, CL, CR, grid)
        r = solver.solve(tol * 10, 1000000, omega=omega)
        stages.append(r.iters)
        ms += r.loop_ms
        g = g * 10
    solver.assemble_3phase(Ds, Df, Dg, CL, CR, grid)
    r = solver.solve(tol, max_iter, omega=omega)
    stages.append(r.iters)
    return dict(deff=r.deff_raw / Df, conv=r.conv, stage_sweeps=stages, path_flag=path, loop_ms=ms + r.loop_ms)


def path_flag_2phase(pix):
    """PathFlag column of BatchSim: FloodFill on Grid = (pixel > 150), cuh:1919-1932."""
    from .solver import flood_fill
    return flood_fill((np.asarray(pix) > 150).astype(np.uint32))[1]


def run_batch(solver, load_image, num_images, Ds, Df, CL, CR, tol, max_iter, rank=0, world=1, dist=None,
              device=None, path_flag=None):
    """Solve images rank, rank+world, ... and gather the table on rank 0.

    solver: a Solver for one image at a time, or a batch Solver (nimg > 1) whose slots are then kept
    full with this rank's images (streaming, every image still stops by its own rule).
    load_image(k) -> uint8 (H, W) pixels of image k (stb-decoded JPEG or synthetic).
    path_flag: callable(pix) -> bool for the PathFlag column (path_flag_2phase), None = -1.
    Returns the (num_images, 9) table on rank 0, None elsewhere.
    """
    mine = shard(num_images, rank, world)
    rows = np.zeros((len(mine), len(COLUMNS)))
    if hasattr(solver, "solve_stream") and getattr(solver, "nimg", 1) > 1:
        # a batch context: keep its slots full with this rank's images (deff_solve_stream)
        stats = []

        def images():
            for k in mine:
                pix = load_image(k)
                stats.append((float(np.count_nonzero(pix < 150)) / pix.size,        # calcPorosity cuh:383-408
                              float(path_flag(pix)) if path_flag is not None else -1.0, pix.size))
                yield pix

        res = solver.solve_stream(images(), Ds, Df, CL, CR, tol, max_iter)
        for slot, (k, r) in enumerate(zip(mine, res)):
            porosity, path, size = stats[slot]
            rows[slot] = (k, porosity, path, r.deff_raw / Df, r.loop_ms / 1000.0, size, r.conv, Ds, Df)
    else:
        for slot, k in enumerate(mine):
            pix = load_image(k)
            porosity = float(np.count_nonzero(pix < 150)) / pix.size            # calcPorosity cuh:383-408
            deff, conv, iters, ms = solve_image(solver, pix, Ds, Df, CL, CR, tol, max_iter)
            path = path_flag(pix) if path_flag is not None else -1.0
            rows[slot] = (k, porosity, float(path), deff, ms / 1000.0, pix.size, conv, Ds, Df)
    if world == 1 or dist is None:
        return rows
    import torch
    # fixed-size exchange: every rank contributes ceil(num_images/world) rows, padded with -1
    per = (num_images + world - 1) // world
    buf = torch.full((per, len(COLUMNS)), -1.0, dtype=torch.float64, device=device or "cpu")
    if len(mine):
        buf[: len(mine)] = torch.from_numpy(rows).to(buf.device)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)
    if rank != 0:
        return None
    table = np.zeros((num_images, len(COLUMNS)))
    for r, part in enumerate(parts):
        part = part.cpu().numpy()
        for slot, k in enumerate(shard(num_images, r, world)):
            table[k] = part[slot]
    return table
