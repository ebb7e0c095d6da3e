"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
committed golden vectors.  Run on the GPU box: python -m pytest tests -m gpu.

Bars (north_star): concentration field <= 1e-6 relative L2 at the same
iteration count, Deff <= 1e-8 relative.  The library is built with
-ffp-contract=off and follows the reference's operation order, so in fact
every comparison below is also asserted bit-exact against the oracle.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELD_TOL = 1e-6     # relative L2, north_star
DEFF_TOL = 1e-8      # relative, north_star
OMEGA = 2.0 / 3.0
KERNELS = ["scalar", "explicit", "matfree", "matfree_tb"]


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def assert_field(got, want):
    """<= 1e-6 relative L2 over the finite cells, NaN/Inf cells in the same places
    (reference semantics for singular rows), and bit-exact on top."""
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got))
    assert rel_l2(got[fin], want[fin]) <= FIELD_TOL
    assert np.array_equal(got, want, equal_nan=True), f"not bit-exact: rel L2 {rel_l2(got[fin], want[fin]):.3e}"


def rand_mask(rng, nx, ny, p=0.5):
    return np.where(rng.random((ny, nx)) < p, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ assembly

@pytest.mark.parametrize("name", ["s8x8", "s16x12", "s33x17"])
def test_assembly_matches_golden_and_oracle(pkg, oracle, small_cases, name):
    pix = small_cases[name + "_pix"]
    Ds, Df, CL, CR = small_cases[name + "_par"]
    ny, nx = pix.shape
    Agold, bgold = small_cases[name + "_A"], small_cases[name + "_b"]
    with pkg.Solver(nx, ny) as s:
        # native path: pixels -> phase codes + lookup tables (what the matrix-free kernel sees)
        s.set_image(pix)
        s.assemble_2phase(Ds, Df, CL, CR)
        A, b = s.get_system()
        assert np.array_equal(A, Agold) and np.array_equal(b, bgold)
        # native path, explicit SoA planes built on the device from the pixels
        s.set_kernel("explicit")
        s.init_linear(CL, CR)
        s.sweeps(0)
        A, b = s.get_system()
        assert np.array_equal(A, Agold) and np.array_equal(b, bgold)
    with pkg.Solver(nx, ny) as s:
        # drop-in path: DiscretizeMatrix2D(D) on the device
        D = oracle.fill_D_2phase(pix, Df, Ds)
        s.assemble_from_D(D, CL, CR)
        A, b = s.get_system()
        assert np.array_equal(A, Agold) and np.array_equal(b, bgold)
        # DiscretizeMatrix2D_ImpSolid
        s.assemble_from_D(oracle.fill_D_2phase(pix, Df, 0.0), CL, CR, grid=small_cases[name + "_grid"])
        A, b = s.get_system()
        assert np.array_equal(A, small_cases[name + "_Aimp"]) and np.array_equal(b, small_cases[name + "_bimp"])
        # host-assembled system round trip (AoS -> SoA -> AoS)
        s.set_system(Agold, bgold, D, CL, CR)
        A, b = s.get_system()
        assert np.array_equal(A, Agold) and np.array_equal(b, bgold)


def test_assembly_mesh_amplification(pkg, oracle):
    rng = np.random.default_rng(3)
    pix = rand_mask(rng, 7, 5)
    for ampX, ampY in [(2, 3), (1, 2), (4, 1)]:
        nx, ny = 7 * ampX, 5 * ampY
        D = oracle.fill_D_2phase(pix, 1.0, 1e-3, ampX, ampY)
        Aor, bor = oracle.discretize(D, 0.0, 1.0)
        with pkg.Solver(nx, ny) as s:
            s.set_image(pix, ampX, ampY)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            A, b = s.get_system()
            assert np.array_equal(A, Aor) and np.array_equal(b, bor)


def test_synth_generator_matches_oracle(pkg, oracle):
    for (nx, ny, img) in [(64, 48, 0), (130, 70, 3)]:
        with pkg.Solver(nx, ny) as s:
            s.synth_image(12345, img)
            assert np.array_equal(s.get_image(), oracle.synth_mask(nx, ny, 12345, img))


def test_launch_plan_is_reported(pkg):
    """deff_get_plan: whole-image tiles for a stack of small images, many short chunks for one big image."""
    with pkg.Solver(128, 128, nimg=64) as s:
        assert s.plan()["tb_T"] == 0                       # nothing planned yet
        s.synth_image(1, 0)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(12)
        p = s.plan()
        assert p["tb_T"] in (4, 6, 8) and p["tb_strips"] == 1 and p["tb_LY"] * p["tb_chunks_per_image"] >= 128
        assert p["tb_blocks"] % 8 == 0 and p["tb_impl"] == 2          # a stack of 1 Mi cells: workgroup tiles
    with pkg.Solver(128, 128, nimg=512) as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        p = s.plan()
        # 512 images of 128^2: each is ONE tall tile (16 waves x 8 rows) that waits for nobody -- resident, whatever their number
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_resident"]) == (2, 16, 8, 1), p
        assert p["tb_strips"] == 1 and p["tb_chunks_per_image"] == 1 and p["tb_LY"] == 128 and p["tb_blocks"] == 512
    with pkg.Solver(512, 512, nimg=32) as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        p = s.plan()
        assert p["tb_impl"] == 1 and p["tb_strips"] == 5 and p["tb_LY"] * p["tb_chunks_per_image"] >= 512   # 8 Mi cells: streaming
    with pkg.Solver(1024, 1024) as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        p = s.plan()
        # ONE image below 4 Mi cells: workgroup tiles, 8 sweeps per pass, every tile resident at once -- the native assembly
        # is link-symmetric, so 12 waves x 5 rows with the matrix rows in registers (k_sweep_wgsym)
        assert p["tb_impl"] == 2 and p["tb_T"] == 8 and (p["tb_NW"], p["tb_R"]) == (12, 5) and p["tb_strips"] == 9
        assert p["tb_resident"] == 1 and p["tb_sym"] == 1
        assert p["tb_strips"] * p["tb_chunks_per_image"] <= 256 and p["tb_LY"] * p["tb_chunks_per_image"] >= 1024
        s.set_tuning("tb_impl", 1)
        s.set_tuning("tb_T", 8)
        s.sweeps(16)
        p = s.plan()
        assert p["tb_impl"] == 1 and p["tb_T"] == 8 and p["tb_strips"] == 9 and p["tb_LY"] >= 8   # 128 + 8 x 112 columns, no halo outside the walls
    with pkg.Solver(4096, 4096) as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(8)
        p = s.plan()
        assert p["tb_impl"] == 1 and p["tb_T"] == 8 and p["tb_strips"] == 37                     # large images stream


def test_synthetic_stack_holds_consecutive_images(pkg, oracle):
    """synth_image(seed, img) on a stack of B images = images img .. img+B-1 of the sequence."""
    nx, ny, B = 70, 48, 3
    with pkg.Solver(nx, ny, nimg=B) as s:
        s.synth_image(12345, 5)
        got = s.get_image()
        for k in range(B):
            assert np.array_equal(got[k * ny:(k + 1) * ny], oracle.synth_mask(nx, ny, 12345, 5 + k))


# -------------------------------------------------------------------- sweeps

@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["s8x8", "s16x12", "s33x17"])
def test_sweeps_match_golden(pkg, oracle, small_cases, name, kernel):
    pix = small_cases[name + "_pix"]
    Ds, Df, CL, CR = small_cases[name + "_par"]
    ny, nx = pix.shape
    with pkg.Solver(nx, ny, kernel=kernel) as s:
        s.set_image(pix)
        s.assemble_2phase(Ds, Df, CL, CR)
        for omega, tag in [(OMEGA, "sor"), (1.0, "v1_")]:
            s.init_linear(CL, CR)
            done = 0
            for k in (1, 2, 100):
                s.sweeps(k - done, omega)
                done = k
                assert_field(s.get_field(), small_cases[f"{name}_{tag}{k}"])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("shape", [(2, 2), (3, 5), (130, 70), (514, 6), (1030, 37), (600, 300)])
def test_sweeps_ragged_shapes_vs_oracle(pkg, oracle, shape, kernel):
    nx, ny = shape
    rng = np.random.default_rng(nx * 1000 + ny)
    pix = rand_mask(rng, nx, ny, 0.6)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want = oracle.sweeps(A, b, x0, 7)
    with pkg.Solver(nx, ny, kernel=kernel) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(7)
        assert_field(s.get_field(), want)
        d, MFL, MFR = s.flux()
        dor, MFLo, MFRo = oracle.flux_deff(want, D, 0.0, 1.0)
        assert d == dor and np.array_equal(MFL, MFLo) and np.array_equal(MFR, MFRo)


@pytest.mark.parametrize("T", [2, 4, 6, 8])
@pytest.mark.parametrize("shape,LY", [((600, 300), 0), ((600, 300), 7), ((1030, 37), 16), ((130, 70), 5),
                                      ((256, 256), 64), ((122, 9), 0), ((2, 64), 0), ((498, 40), 0), ((250, 33), 11)])
def test_temporal_blocking_vs_oracle(pkg, oracle, shape, LY, T):
    """T sweeps per pass, strips and chunks of every raggedness, sweep counts that are
    not multiples of T (the remainder runs on the single-sweep kernel)."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 7 + ny * 13 + T)
    pix = rand_mask(rng, nx, ny, 0.55)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    nsw = 3 * T + 3
    want = oracle.sweeps(A, b, x0, nsw)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 1)                             # the streaming form (workgroup tiles: tests/test_gpu_wgtile.py)
        s.set_tuning("tb_T", T)
        s.set_tuning("tb_LY", LY)
        s.set_tuning("tb_wall_halo", (nx + ny + T) % 3)        # all three strip placements get exercised
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(nsw)
        launches, per = s.last_launches()
        assert s.kernel_in_use() == "matfree_tb" and per == T and launches == nsw // T + nsw % T
        assert_field(s.get_field(), want)


def test_temporal_blocking_zero_diffusivity_guard(pkg, oracle, img00000):
    """Ds = 0: links are -0.0 and singular cells go NaN; the guarded variant keeps the
    reference's skip semantics, NaNs appear in exactly the same cells."""
    D = oracle.fill_D_2phase(img00000, 1.0, 0.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0)
        want = oracle.sweeps(A, b, oracle.linear_guess(128, 128, 0.0, 1.0), 13)
    assert np.isnan(want).any() and np.isfinite(want).any()
    with pkg.Solver(128, 128, kernel="matfree_tb") as s:
        s.set_image(img00000)
        s.assemble_2phase(0.0, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(13)
        assert_field(s.get_field(), want)


def test_img00000_3phase_as_shipped(pkg, oracle, recorded, img00000):
    """The configuration the reference's shipped input.txt selects: 3 phases, Ds = 0, Dg = 1237500,
    DCG continuation (6 JacobiGPUPreCond stages + JacobiGPU), FloodFill, ImpSolid rows.  Every
    recorded number of the reference is reproduced: stage sweep counts, Deff, conv."""
    from effectivediffusivityfvm_amd import batch
    rec = recorded["img00000_3phase_as_shipped"]
    o = rec["options"]
    with pkg.Solver(128, 128) as s:
        r = batch.solve_image_3phase(s, img00000, o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"],
                                     o["max_iter"])
        field = s.get_field()
        assert s.kernel_in_use() == "matfree_tb"        # ImpSolid rows harvested into a dictionary
    assert r["stage_sweeps"] == rec["stage_sweeps"]
    assert abs(r["deff"] - rec["deff"]) <= DEFF_TOL * rec["deff"]
    assert r["deff"] == rec["deff"] and r["conv"] == rec["conv"]
    with np.errstate(all="ignore"):
        want = oracle.solve_3phase(img00000, o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"], o["max_iter"])
    assert_field(field, want["field"])
    solid = img00000 > 200
    assert np.all(np.abs(field[solid]) < 1e-300)          # impermeable solid decays to 0 (x <- x/3 per sweep)


def test_img00000_3phase_as_shipped_over_row_slabs(pkg, oracle, recorded, img00000):
    """The same as-shipped 3-phase run with the image split into three row slabs: every slab harvests
    its own row dictionary; stage sweeps, Deff, conv and the field are those of the one-GPU run."""
    from effectivediffusivityfvm_amd import batch
    rec = recorded["img00000_3phase_as_shipped"]
    o = rec["options"]
    with pkg.SlabGroup(128, 128, [0, 0, 0]) as g:
        r = batch.solve_image_3phase(g, img00000, o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"],
                                     o["max_iter"])
        field = g.get_field()
    assert r["stage_sweeps"] == rec["stage_sweeps"]
    assert r["deff"] == rec["deff"] and r["conv"] == rec["conv"]
    with np.errstate(all="ignore"):
        want = oracle.solve_3phase(img00000, o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"], o["max_iter"])
    assert_field(field, want["field"])


def test_3phase_assembly_with_grid_and_amplification(pkg, oracle):
    rng = np.random.default_rng(9)
    pix = rng.choice(np.array([0, 30, 120, 150, 199, 201, 255], dtype=np.uint8), size=(9, 11))
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.2, 0.9, grid=grid)
    with pkg.Solver(11, 9) as s:
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 50.0, 0.2, 0.9, grid)
        Ag, bg = s.get_system()
    assert np.array_equal(Ag, A, equal_nan=True) and np.array_equal(bg, b, equal_nan=True)
    D2 = oracle.fill_D_3phase(pix, 1.0, 0.5, 50.0, ampX=2, ampY=3)
    A2, b2 = oracle.discretize(D2, 0.0, 1.0)
    with pkg.Solver(22, 27) as s:
        s.set_image(pix, 2, 3)
        s.assemble_3phase(0.5, 1.0, 50.0, 0.0, 1.0, None)
        Ag, bg = s.get_system()
    assert np.array_equal(Ag, A2) and np.array_equal(bg, b2)


# ------------------------------------------------------------------- batches

@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("shape,B", [((130, 70), 5), ((64, 9), 3), ((33, 17), 4), ((256, 128), 2)])
def test_batch_sweeps_equal_single_image_runs(pkg, oracle, shape, B, kernel):
    """A stacked batch is swept as one domain; every image must come out exactly as if it had
    been swept alone (the zero-flux top/bottom walls keep the images uncoupled)."""
    nx, ny = shape
    rng = np.random.default_rng(B * 100 + nx)
    pix = np.stack([rand_mask(rng, nx, ny, 0.4 + 0.05 * k) for k in range(B)])
    x0 = rng.random((B, ny, nx))
    with pkg.Solver(nx, ny, kernel=kernel, nimg=B) as s:
        s.set_tuning("tb_LY", 16)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        A, b = s.get_system()
        s.set_field(x0.reshape(B * ny, nx))
        s.sweeps(11)
        got = s.get_field().reshape(B, ny, nx)
        deffs, MFL, MFR = s.flux()
    for k in range(B):
        D = oracle.fill_D_2phase(pix[k], 1.0, 1e-3)
        Ao, bo = oracle.discretize(D, 0.0, 1.0)
        n = nx * ny
        assert np.array_equal(A[k * n:(k + 1) * n], Ao) and np.array_equal(b[k * n:(k + 1) * n], bo)
        want = oracle.sweeps(Ao, bo, x0[k], 11)
        assert_field(got[k], want)
        d, l, r = oracle.flux_deff(want, D, 0.0, 1.0)
        assert deffs[k] == d and np.array_equal(MFL[k * ny:(k + 1) * ny], l)


@pytest.mark.parametrize("kernel", ["matfree_tb", "explicit", "scalar"])
def test_batch_solve_each_image_stops_by_its_own_rule(pkg, oracle, kernel):
    """Images of one batch converge after different numbers of checks; each must report the
    sweep count, Deff, conv and field of a one-image run of the reference loop."""
    nx, ny, B = 64, 48, 5
    rng = np.random.default_rng(42)
    pix = np.stack([rand_mask(rng, nx, ny, p) for p in (0.3, 0.5, 0.7, 0.9, 0.5)])
    pix[3] = 255
    pix[3, :10, :] = 0                      # parallel stripes: exact after the first interval
    with pkg.Solver(nx, ny, kernel=kernel, nimg=B) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(2e-3, 5000, check_every=100)
        got = s.get_field().reshape(B, ny, nx)
        # warm start of the whole batch from the frozen fields
        res2 = s.solve(5e-4, 300, check_every=100)
        got2 = s.get_field().reshape(B, ny, nx)
    its = set()
    for k in range(B):
        D = oracle.fill_D_2phase(pix[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0,
                                                    2e-3, 5000, check_every=100)
        its.add(it)
        r = res[k]
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv), (k, r, it, deff, conv)
        assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)
        assert_field(got[k], x)
        it2, deff2, conv2, x2, _, _ = oracle.jacobi(A, b, x, D, 0.0, 1.0, 5e-4, 300, check_every=100)
        assert (res2[k].iters, res2[k].deff_raw, res2[k].conv) == (it2, deff2, conv2)
        assert_field(got2[k], x2)
    assert len(its) >= 3, its                # the batch really did split up


@pytest.mark.parametrize("slots,max_iter", [(3, 5000), (4, 730), (2, 1), (16, 5000)])
def test_streaming_batch_refills_slots(pkg, oracle, slots, max_iter):
    """11 images through `slots` slots: finished images are replaced on the fly, and every image
    still reports exactly what a one-image run of the reference loop gives (sweep count, Deff,
    conv, field), including images that run into MAX_ITER between checks."""
    nx, ny = 64, 48
    rng = np.random.default_rng(2024)
    imgs = [rand_mask(rng, nx, ny, p) for p in (0.3, 0.5, 0.7, 0.9, 0.5, 0.2, 0.8, 0.6, 0.4, 0.55, 0.35)]
    imgs[3][:] = 255
    imgs[3][:10, :] = 0                      # parallel stripes: stops at its second check
    with pkg.Solver(nx, ny, nimg=slots) as s:
        res = s.solve_stream(imgs, 1e-2, 1.0, 0.0, 1.0, 2e-3, max_iter, check_every=100, want_fields=True)
    assert len(res) == len(imgs)
    counts = set()
    for k, pix in enumerate(imgs):
        D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 2e-3,
                                                max_iter, check_every=100)
        counts.add(it)
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv), (k, res[k], it, deff, conv)
        assert_field(res[k].field, x)
    if max_iter == 5000:
        assert len(counts) >= 4


def test_run_batch_table_streaming_equals_one_at_a_time(pkg, oracle):
    """batch.run_batch (BatchSim's NumImg x 9 table) through a streaming batch context and through a
    one-image context: same table, and its Deff column equals the oracle's."""
    from effectivediffusivityfvm_amd import batch
    nx, ny, N = 48, 40, 9

    def load(k):
        return oracle.synth_mask(nx, ny, 999, k)

    with pkg.Solver(nx, ny, nimg=4) as s4, pkg.Solver(nx, ny) as s1:
        t4 = batch.run_batch(s4, load, N, 1e-2, 2.0, 0.0, 1.0, 1e-3, 20000, path_flag=batch.path_flag_2phase)
        t1 = batch.run_batch(s1, load, N, 1e-2, 2.0, 0.0, 1.0, 1e-3, 20000, path_flag=batch.path_flag_2phase)
    keep = [i for i, c in enumerate(batch.COLUMNS) if c != "Time"]
    assert np.array_equal(t4[:, keep], t1[:, keep])
    for k in range(N):
        D = oracle.fill_D_2phase(load(k), 2.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, _, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 20000)
        assert t4[k, 3] == deff / 2.0 and t4[k, 6] == conv and t4[k, 0] == k


def test_batch_synthetic_first_check_at_1024(pkg, oracle, recorded):
    """4 stacked 1024^2 synthetic images = images 0..3 of the generator; image 0 must give the
    reference's recorded first-check Deff, the others the oracle's."""
    B, n = 4, 1024
    with pkg.Solver(n, n, nimg=B) as s:
        s.synth_image(12345, 0)
        pix = s.get_image().reshape(B, n, n)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-6, 1)
    assert res[0].deff_raw == recorded["synthetic_first_check_deff"]["1024"]
    for k in range(B):
        assert np.array_equal(pix[k], oracle.synth_mask(n, n, 12345, k))
    D = oracle.fill_D_2phase(pix[2], 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    d, _, _ = oracle.flux_deff(oracle.sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 1), D, 0.0, 1.0)
    assert res[2].deff_raw == d


# ----------------------------------------------------------------- row slabs

@pytest.mark.parametrize("overlap", [2, 0])
@pytest.mark.parametrize("nslabs", [2, 3, 4])
@pytest.mark.parametrize("T", [0, 2, 8])
def test_row_slabs_equal_single_domain(pkg, oracle, nslabs, T, overlap):
    """One image over several slabs with halo exchange once per blocked pass (all slabs on
    device 0 here): the assembled field is bit-identical to the oracle / one-context field, the
    fluxes and Deff too (they are summed in global row order)."""
    nx, NY = 384, 203
    rng = np.random.default_rng(nslabs * 10 + T)
    pix = rand_mask(rng, nx, NY, 0.55)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((NY, nx))
    want = oracle.sweeps(A, b, x0, 29)
    with pkg.SlabGroup(nx, NY, [0] * nslabs) as g:
        first, count = g.layout()
        assert first[0] == 0 and sum(count) == NY and all(c >= 8 for c in count)
        # overlap 2: per pass the two boundary bands, the exchange on a second stream, the interior meanwhile (what slabs
        # of >= 16 Mi cells do by default); overlap 0: pass, exchange, pass on one stream.  Same bits either way.
        g.set_tuning("slab_overlap", overlap)
        g.set_tuning("tb_T", T)
        g.set_image(pix)
        g.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        g.set_field(x0)
        g.sweeps(29)                       # passes of T plus single-sweep passes, exchange after each
        assert_field(g.get_field(), want)
        d, MFL, MFR = g.flux()
        dor, MFLo, MFRo = oracle.flux_deff(want, D, 0.0, 1.0)
        assert d == dor and np.array_equal(MFL, MFLo) and np.array_equal(MFR, MFRo)


def test_row_slabs_solve_matches_single_context(pkg, oracle):
    nx, NY = 256, 160
    rng = np.random.default_rng(77)
    pix = rand_mask(rng, nx, NY, 0.6)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, oracle.linear_guess(nx, NY, 0.0, 1.0), D, 0.0, 1.0, 1e-3,
                                                3000, check_every=100)
    with pkg.SlabGroup(nx, NY, [0, 0, 0]) as g:
        g.set_image(pix)
        g.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        g.init_linear(0.0, 1.0)
        r = g.solve(1e-3, 3000, check_every=100)
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert_field(g.get_field(), x)
        assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)


def test_row_slabs_4096_equals_one_gpu(pkg, recorded):
    """BASELINE config #4's check at a size that fits one GPU: 4 slabs of a 4096^2 synthetic image
    against the one-context run, bit for bit, plus the reference's recorded first-check Deff."""
    n = 4096
    with pkg.Solver(n, n) as s:
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(37)
        ref = s.get_field()
    with pkg.SlabGroup(n, n, [0, 0, 0, 0]) as g:
        g.synth_image(12345, 0)
        g.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        g.init_linear(0.0, 1.0)
        r = g.solve(1e-6, 1)
        assert r.deff_raw == recorded["synthetic_first_check_deff"]["4096"]
        g.sweeps(36)
        assert np.array_equal(g.get_field(), ref)


def test_slabs_of_an_odd_width_image(pkg, oracle):
    """Row slabs with padded device rows: 3 slabs of a 131-column image, full solve."""
    nx, NY = 131, 90
    rng = np.random.default_rng(31)
    pix = rand_mask(rng, nx, NY, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((NY, nx))
    it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-3, 2000, check_every=100)
    with pkg.SlabGroup(nx, NY, [0, 0, 0]) as g:
        g.set_image(pix)
        g.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        g.set_field(x0)
        assert np.array_equal(g.get_field(), x0)
        r = g.solve(1e-3, 2000, check_every=100)
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)
        assert_field(g.get_field(), x)
    with pkg.SlabGroup(nx, NY, [0, 0]) as g:                     # the generator through slab windows
        g.synth_image(12345, 2)
        g.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        g.init_linear(0.0, 1.0)
        g.sweeps(19)
        pix2 = oracle.synth_mask(nx, NY, 12345, 2)
        D2 = oracle.fill_D_2phase(pix2, 1.0, 1e-2)
        A2, b2 = oracle.discretize(D2, 0.0, 1.0)
        assert_field(g.get_field(), oracle.sweeps(A2, b2, oracle.linear_guess(nx, NY, 0.0, 1.0), 19))


def test_rccl_slab_single_rank(pkg, oracle):
    """The process-per-GPU transport with a communicator of one rank (all a one-GPU box allows):
    RCCL init, the all-gather of the fluxes, the solve loop; no neighbour to exchange with."""
    nx, NY = 128, 96
    rng = np.random.default_rng(3)
    pix = rand_mask(rng, nx, NY, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, oracle.linear_guess(nx, NY, 0.0, 1.0), D, 0.0, 1.0, 1e-3,
                                                2000, check_every=100)
    with pkg.SlabRank(nx, NY, 0, 1, pkg.rccl_unique_id()) as s:
        assert s.layout() == (0, NY) and s.window() == (0, NY)
        s.set_image(pix)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-3, 2000, check_every=100)
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert_field(s.get_field(), x)
        assert np.array_equal(r.MFL, MFL)


def _rccl_slab_worker(rank, world, idfile, nx, NY, out_dir):
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import time
    import effectivediffusivityfvm_amd as pkg
    if rank == 0:
        uid = pkg.rccl_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    pix = np.load(os.path.join(out_dir, "pix.npy"))
    with pkg.SlabRank(nx, NY, rank, world, uid, device=rank) as s:
        s.set_tuning("slab_overlap", 2)           # grouped ncclSend/ncclRecv on the copy stream, interior sweeps meanwhile
        s.set_image(pix)
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-3, 2000, check_every=100)
        np.save(os.path.join(out_dir, f"x{rank}.npy"), s.get_field())
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([r.iters, r.deff_raw, r.conv]))


def test_rccl_slabs_two_ranks(pkg, oracle, tmp_path):
    """Two processes, two GPUs, RCCL halo exchange: needs a multi-GPU box (skipped on one GPU)."""
    import subprocess
    n = int(subprocess.run(["python3", "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True).stdout.strip() or 0)
    if n < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    nx, NY = 256, 200
    rng = np.random.default_rng(5)
    pix = rand_mask(rng, nx, NY, 0.5)
    np.save(tmp_path / "pix.npy", pix)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, NY, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 2000,
                                            check_every=100)
    from conftest import spawn_with_timeout
    spawn_with_timeout(_rccl_slab_worker, (2, str(tmp_path / "id"), nx, NY, str(tmp_path)), 2, timeout_s=240)
    got = np.concatenate([np.load(tmp_path / "x0.npy"), np.load(tmp_path / "x1.npy")])
    assert_field(got, x)
    for k in range(2):
        assert tuple(np.load(tmp_path / f"r{k}.npy")) == (it, deff, conv)


def _gloo_slab_worker(rank, world, port, nx, NY, out_dir, three_phase=False):
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import effectivediffusivityfvm_amd as pkg
    from effectivediffusivityfvm_amd.solver import TorchDistTransport
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pix = np.load(os.path.join(out_dir, "pix.npy"))
    with pkg.SlabRank(nx, NY, rank, world, device=0, transport=TorchDistTransport()) as s:
        # 2-phase run: the exchange overlapped with the interior (bands first, second stream), as slabs of >= 16 Mi cells
        # do by default; 3-phase run: pass, exchange, pass on one stream
        s.set_tuning("slab_overlap", 0 if three_phase else 2)
        s.set_image(pix)
        if three_phase:
            s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid_full=np.load(os.path.join(out_dir, "grid.npy")))
        else:
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-3, 2000, check_every=100)
        np.save(os.path.join(out_dir, f"x{rank}.npy"), s.get_field())
        np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([r.iters, r.deff_raw, r.conv]))
        np.save(os.path.join(out_dir, f"mfl{rank}.npy"), r.MFL)
    dist.destroy_process_group()


def test_slab_ranks_three_processes_one_gpu(pkg, oracle, tmp_path):
    """The process-per-GPU slab loop across real process boundaries: three processes share this
    box's one GPU, the halo blocks and fluxes travel host-staged over gloo (the custom transport).
    Everything but the RCCL calls themselves is the code a multi-GPU run executes; uneven slabs
    (NY=203 over 3), a middle rank with two neighbours."""
    import socket
    import torch.multiprocessing as mp
    nx, NY, world = 255, 203, 3                      # odd width: padded device rows, host rows of 255
    rng = np.random.default_rng(5)
    pix = rand_mask(rng, nx, NY, 0.5)
    np.save(tmp_path / "pix.npy", pix)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, MFL, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, NY, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 2000,
                                              check_every=100)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    from conftest import spawn_with_timeout
    spawn_with_timeout(_gloo_slab_worker, (world, port, nx, NY, str(tmp_path)), world, timeout_s=240)
    got = np.concatenate([np.load(tmp_path / f"x{k}.npy") for k in range(world)])
    assert_field(got, x)
    for k in range(world):
        assert tuple(np.load(tmp_path / f"r{k}.npy")) == (it, deff, conv)
        assert np.array_equal(np.load(tmp_path / f"mfl{k}.npy"), MFL)


# ---- odd mesh widths: arrays padded to an even pitch, every fast kernel available ---------------

@pytest.mark.parametrize("nx,ny", [(97, 41), (1001, 333), (3, 5), (129, 64)])
def test_odd_width_runs_on_the_blocked_kernel(pkg, oracle, nx, ny):
    """An odd nx is padded by one column of cells outside the mesh; geometry (dx, wall column, linear
    guess) stays that of the true width.  Full solve, fluxes, field, system export."""
    rng = np.random.default_rng(nx + ny)
    pix = rand_mask(rng, nx, ny, 0.55)
    CL, CR = 0.2, 1.1
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, CL, CR)
    x0 = oracle.linear_guess(nx, ny, CL, CR)
    it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, x0, D, CL, CR, 1e-4, 3000, check_every=100)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-2, 1.0, CL, CR)
        A2, b2 = s.get_system()
        assert np.array_equal(A2, A) and np.array_equal(b2, b)
        s.init_linear(CL, CR)
        assert np.array_equal(s.get_field(), x0)
        r = s.solve(1e-4, 3000, check_every=100)
        assert s.kernel_in_use() == ("matfree_tb" if ny >= 8 else "matfree")
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)
        assert_field(s.get_field(), x)
    # the reference's own arrays in (drop-in route), explicit kernel and harvested dictionary
    for kernel in ("explicit", "auto"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_system(A, b, D, CL, CR)
            A3, b3 = s.get_system()
            assert np.array_equal(A3, A) and np.array_equal(b3, b)
            s.set_field(x0)
            r = s.solve(1e-4, 3000, check_every=100)
            assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
            assert_field(s.get_field(), x)


def test_odd_width_three_phase_batch_and_stream(pkg, oracle):
    nx, ny = 75, 40
    rng = np.random.default_rng(9)
    # 3-phase with Grid through assemble_3phase and assemble_from_D
    pix = rng.choice(np.array([0, 120, 255], dtype=np.uint8), size=(ny, nx), p=[0.3, 0.4, 0.3])
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    grid = (pix > 200).astype(np.uint32)
    A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
    ref = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 23)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid=grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(23)
        assert_field(s.get_field(), ref)
        s.assemble_from_D(D, 0.0, 1.0, grid=grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(23)
        assert_field(s.get_field(), ref)
    # a stack of 5 odd-width images: every image equal to its one-image run
    imgs = [rand_mask(rng, nx, ny, 0.5) for _ in range(5)]
    want = []
    for im in imgs:
        Dk = oracle.fill_D_2phase(im, 1.0, 1e-2)
        Ak, bk = oracle.discretize(Dk, 0.0, 1.0)
        want.append(oracle.jacobi(Ak, bk, oracle.linear_guess(nx, ny, 0.0, 1.0), Dk, 0.0, 1.0, 1e-3, 2000, check_every=100))
    with pkg.Solver(nx, ny, nimg=5) as s:
        s.set_image(np.concatenate(imgs))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-3, 2000, check_every=100)
        x = s.get_field()
        for k, (it, deff, conv, xk, _, _) in enumerate(want):
            assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
            assert_field(x[k * ny:(k + 1) * ny], xk)
    with pkg.Solver(nx, ny, nimg=2) as s:
        got = s.solve_stream(imgs, 1e-2, 1.0, 0.0, 1.0, 1e-3, 2000, check_every=100, want_fields=True)
        for k, (it, deff, conv, xk, _, _) in enumerate(want):
            r = got[k]
            assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
            assert_field(r.field, xk)


# ---- randomised small cases -----------------------------------------------------------------

def test_random_small_cases_all_paths(pkg, oracle):
    """240 seeded random configurations -- mesh 2..150 x 2..90 (odd and even), 1-3 stacked images,
    every kernel, every T, both arithmetics, 2 phases / 3 phases with a flood-filled Grid, random
    diffusivities and wall values, 1..40 sweeps -- each compared bit for bit with the oracle."""
    rng = np.random.default_rng(20261004)
    for case in range(240):
        nx, ny = int(rng.integers(2, 151)), int(rng.integers(2, 91))
        nimg = int(rng.choice([1, 1, 2, 3]))
        kernel = str(rng.choice(["auto", "matfree_tb", "matfree", "explicit", "scalar"]))
        T = int(rng.choice([0, 1, 2, 4, 6, 8]))
        fma = int(rng.integers(0, 2))
        three = bool(rng.integers(0, 2))
        nsw = int(rng.integers(1, 41))
        CL, CR = float(rng.uniform(-1, 1)), float(rng.uniform(-1, 2))
        flav = "fma" if fma else None
        if three:
            Ds, Df, Dg = 0.0, float(rng.uniform(0.5, 2)), float(rng.uniform(5, 500))
        else:
            Ds, Df, Dg = float(10 ** rng.uniform(-4, 0)), float(rng.uniform(0.5, 2)), 0.0
        imgs, refs = [], []
        for _ in range(nimg):
            if three:
                pix = rng.choice(np.array([0, 120, 255], dtype=np.uint8), size=(ny, nx), p=[0.3, 0.4, 0.3])
                D = oracle.fill_D_3phase(pix, Df, Ds, Dg)
                grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
                with np.errstate(all="ignore"):
                    A, b = oracle.discretize(D, CL, CR, grid=grid)
            else:
                pix = rand_mask(rng, nx, ny, float(rng.uniform(0.2, 0.8)))
                D = oracle.fill_D_2phase(pix, Df, Ds)
                A, b = oracle.discretize(D, CL, CR)
                grid = None
            imgs.append((pix, grid))
            with np.errstate(all="ignore"):
                refs.append(oracle.sweeps(A, b, oracle.linear_guess(nx, ny, CL, CR, flavour=flav), nsw, flavour=flav))
        tag = f"case {case}: {nx}x{ny} x{nimg} {kernel} T={T} fma={fma} {'3' if three else '2'}-phase {nsw} sweeps"
        with pkg.Solver(nx, ny, kernel=kernel, nimg=nimg) as s:
            s.set_tuning("fma", fma)
            if T:
                s.set_tuning("tb_T", T)
            s.set_image(np.concatenate([p for p, _ in imgs]))
            if three:
                s.assemble_3phase(Ds, Df, Dg, CL, CR, grid=np.concatenate([g for _, g in imgs]))
            else:
                s.assemble_2phase(Ds, Df, CL, CR)
            s.init_linear(CL, CR)
            s.sweeps(nsw)
            got = s.get_field()
        for k in range(nimg):
            a, w = got[k * ny:(k + 1) * ny], refs[k]
            assert a.shape == w.shape, tag
            same = (a == w) | (np.isnan(a) & np.isnan(w))
            assert same.all(), f"{tag}: image {k}, {np.count_nonzero(~same)} cells differ"


def test_random_stopping_rules_batch_and_stream(pkg, oracle):
    """80 seeded random solves: check interval, MaxIter and tolerance drawn at random (including the
    degenerate ones: tolerance above 100 = no sweep, MaxIter below the first check interval, checks
    every sweep), one image / a stack / a stream through 1-3 slots; every image's sweep count, Deff,
    conv and field equal the oracle's one-image loop."""
    rng = np.random.default_rng(777)
    for case in range(80):
        nx, ny = int(rng.integers(8, 81)), int(rng.integers(8, 61))
        nimg = int(rng.integers(1, 5))
        check_every = int(rng.choice([1, 2, 3, 10, 37, 50]))
        max_iter = int(rng.integers(1, 301))
        tol = float(rng.choice([0.0, 1e-2, 1e-3, 1e-5, 150.0]))
        Ds = float(10 ** rng.uniform(-3, 0))
        CL, CR = 0.0, float(rng.uniform(0.5, 2))
        imgs = [rand_mask(rng, nx, ny, float(rng.uniform(0.3, 0.7))) for _ in range(nimg)]
        want = []
        for pix in imgs:
            D = oracle.fill_D_2phase(pix, 1.0, Ds)
            A, b = oracle.discretize(D, CL, CR)
            want.append(oracle.jacobi(A, b, oracle.linear_guess(nx, ny, CL, CR), D, CL, CR, tol, max_iter,
                                      check_every=check_every))
        tag = f"case {case}: {nx}x{ny} x{nimg} C={check_every} max={max_iter} tol={tol}"
        with pkg.Solver(nx, ny, nimg=nimg) as s:
            s.set_image(np.concatenate(imgs))
            s.assemble_2phase(Ds, 1.0, CL, CR)
            s.init_linear(CL, CR)
            res = s.solve(tol, max_iter, check_every=check_every)
            res = [res] if nimg == 1 else res
            x = s.get_field()
            for k, (it, deff, conv, xk, _, _) in enumerate(want):
                assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv), f"{tag}: batch image {k}"
                assert_field(x[k * ny:(k + 1) * ny], xk)
        slots = int(rng.integers(1, 4))
        with pkg.Solver(nx, ny, nimg=slots) as s:
            got = s.solve_stream(imgs, Ds, 1.0, CL, CR, tol, max_iter, check_every=check_every, want_fields=True)
            for k, (it, deff, conv, xk, _, _) in enumerate(want):
                assert (got[k].iters, got[k].deff_raw, got[k].conv) == (it, deff, conv), f"{tag}: stream image {k} of {slots} slots"
                assert_field(got[k].field, xk)


def test_random_row_slabs(pkg, oracle):
    """50 seeded random slab decompositions (1-5 slabs, odd / even widths, every T, both arithmetics,
    2 or 3 phases): sweeps and a short solve equal the oracle bit for bit."""
    rng = np.random.default_rng(4242)
    for case in range(50):
        nslabs = int(rng.integers(1, 6))
        nx, NY = int(rng.integers(2, 300)), int(rng.integers(8 * nslabs, 8 * nslabs + 120))
        T = int(rng.choice([0, 1, 2, 4, 6, 8]))
        fma = int(rng.integers(0, 2))
        three = bool(rng.integers(0, 2))
        nsw = int(rng.integers(1, 50))
        flav = "fma" if fma else None
        if three:
            pix = rng.choice(np.array([0, 120, 255], dtype=np.uint8), size=(NY, nx), p=[0.3, 0.4, 0.3])
            D = oracle.fill_D_3phase(pix, 1.0, 0.0, 40.0)
            grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
            with np.errstate(all="ignore"):
                A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
        else:
            pix = rand_mask(rng, nx, NY, 0.5)
            D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
            A, b = oracle.discretize(D, 0.0, 1.0)
        x0 = oracle.linear_guess(nx, NY, 0.0, 1.0, flavour=flav)
        with np.errstate(all="ignore"):
            ref = oracle.sweeps(A, b, x0, nsw, flavour=flav)
            it, deff, conv, xs, MFL, MFR = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-3, 150, check_every=20, flavour=flav)
        tag = f"case {case}: {nx}x{NY} in {nslabs} slabs T={T} fma={fma} {'3' if three else '2'}-phase"
        with pkg.SlabGroup(nx, NY, [0] * nslabs) as g:
            g.set_tuning("fma", fma)
            if T:
                g.set_tuning("tb_T", T)
            g.set_image(pix)
            for _ in range(2):                       # sweeps, then a solve from a fresh guess
                if three:
                    g.assemble_3phase(0.0, 1.0, 40.0, 0.0, 1.0, grid=grid)
                else:
                    g.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
                g.init_linear(0.0, 1.0)
                if _ == 0:
                    g.sweeps(nsw)
                    a = g.get_field()
                    same = (a == ref) | (np.isnan(a) & np.isnan(ref))
                    assert same.all(), f"{tag}: {np.count_nonzero(~same)} cells differ after {nsw} sweeps"
                else:
                    r = g.solve(1e-3, 150, check_every=20)
                    assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv), tag
                    assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR), tag
                    a = g.get_field()
                    assert ((a == xs) | (np.isnan(a) & np.isnan(xs))).all(), tag


# ---- contracted arithmetic (opt-in): the oracle's "fma" build is the checker ------------------

@pytest.mark.parametrize("kernel,nx,ny", [("explicit", 96, 64), ("scalar", 97, 41), ("matfree", 96, 64),
                                          ("matfree", 97, 41), ("matfree_tb", 260, 70)])
def test_fma_mode_matches_contracted_oracle(pkg, oracle, kernel, nx, ny):
    """deff_set_tuning("fma", 1): every sweep kernel (and the linear guess) is bit-identical to the
    oracle built with contraction allowed -- and differs from the default arithmetic, so the switch
    is really exercised."""
    rng = np.random.default_rng(21)
    pix = rand_mask(rng, nx, ny, 0.45)
    CL, CR = 0.1, 0.93
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, CL, CR)
    x0 = oracle.linear_guess(nx, ny, CL, CR, flavour="fma")
    nsw = 37
    ref = oracle.sweeps(A, b, x0, nsw, flavour="fma")
    plain = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, CL, CR), nsw)
    assert not np.array_equal(ref, plain)
    with pkg.Solver(nx, ny, kernel=kernel) as s:
        s.set_tuning("fma", 1)
        s.set_image(pix)
        s.assemble_2phase(1e-2, 1.0, CL, CR)
        s.init_linear(CL, CR)
        assert np.array_equal(s.get_field(), x0)
        s.sweeps(nsw)
        assert s.kernel_in_use() == kernel
        assert_field(s.get_field(), ref)
        s.set_tuning("fma", 0)                       # and back: the default arithmetic again
        s.init_linear(CL, CR)
        s.sweeps(nsw)
        assert_field(s.get_field(), plain)


def test_fma_mode_three_phase_guarded_rows(pkg, oracle):
    """Zero-diffusivity solid (links -0.0, NaN-free thanks to the guard) through the dictionary and the
    temporally blocked kernel in contracted arithmetic."""
    rng = np.random.default_rng(22)
    nx, ny = 192, 80
    pix = rng.choice(np.array([0, 120, 255], dtype=np.uint8), size=(ny, nx), p=[0.3, 0.4, 0.3])
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    grid = (pix > 200).astype(np.uint32)
    A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0, flavour="fma")
    ref = oracle.sweeps(A, b, x0, 29, flavour="fma")
    for kernel in ("auto", "explicit"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_tuning("fma", 1)
            s.set_image(pix)
            s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid=grid)
            s.init_linear(0.0, 1.0)
            s.sweeps(29)
            assert s.kernel_in_use() == ("matfree_tb" if kernel == "auto" else "explicit")
            assert_field(s.get_field(), ref)


def test_fma_mode_through_stream_and_slabs(pkg, oracle):
    """The arithmetic switch reaches the streaming batch (its own linear guesses) and the row slabs."""
    nx, ny = 130, 72
    rng = np.random.default_rng(41)
    imgs = [rand_mask(rng, nx, ny, 0.5) for _ in range(4)]
    want = []
    for im in imgs:
        D = oracle.fill_D_2phase(im, 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.1, 0.9)
        want.append(oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.1, 0.9, flavour="fma"), D, 0.1, 0.9, 1e-3, 2000,
                                  check_every=100, flavour="fma"))
    with pkg.Solver(nx, ny, nimg=2) as s:
        s.set_tuning("fma", 1)
        got = s.solve_stream(imgs, 1e-2, 1.0, 0.1, 0.9, 1e-3, 2000, check_every=100, want_fields=True)
        for k, (it, deff, conv, xk, _, _) in enumerate(want):
            assert (got[k].iters, got[k].deff_raw, got[k].conv) == (it, deff, conv)
            assert_field(got[k].field, xk)
    with pkg.SlabGroup(nx, ny, [0, 0, 0]) as g:
        g.set_tuning("fma", 1)
        g.set_image(imgs[0])
        g.assemble_2phase(1e-2, 1.0, 0.1, 0.9)
        g.init_linear(0.1, 0.9)
        r = g.solve(1e-3, 2000, check_every=100)
        it, deff, conv, xk, _, _ = want[0]
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert_field(g.get_field(), xk)


def test_fma_mode_config1_gives_the_surveys_primary_value(pkg, oracle, recorded, img00000):
    """Config #1 end to end in contracted arithmetic: 110 001 sweeps and the Deff / conv the survey
    recorded first for the reference (build a); the default arithmetic gives build b
    (test_img00000_config1_end_to_end)."""
    rec = recorded["img00000_2phase_batch"]
    with pkg.Solver(128, 128) as s:
        s.set_tuning("fma", 1)
        s.set_image(img00000)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-6, 500000)
    assert r.iters == rec["iters"]
    assert r.deff_raw == rec["deff_build_a"]
    assert r.conv == rec["conv_build_a"]


def test_slab_ranks_three_phase_three_processes(pkg, oracle, tmp_path):
    """3-phase (impermeable solid, flood-filled Grid) through the per-rank slab path: three
    processes, custom transport, each rank assembling its window of the Grid."""
    import socket
    import torch.multiprocessing as mp
    nx, NY, world = 130, 96, 3
    rng = np.random.default_rng(6)
    pix = rng.choice(np.array([0, 120, 255], dtype=np.uint8), size=(NY, nx), p=[0.35, 0.4, 0.25])
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    np.save(tmp_path / "pix.npy", pix)
    np.save(tmp_path / "grid.npy", grid)
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
        it, deff, conv, x, MFL, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, NY, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 2000,
                                                  check_every=100)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    from conftest import spawn_with_timeout
    spawn_with_timeout(_gloo_slab_worker, (world, port, nx, NY, str(tmp_path), True), world, timeout_s=240)
    got = np.concatenate([np.load(tmp_path / f"x{k}.npy") for k in range(world)])
    assert_field(got, x)
    for k in range(world):
        assert tuple(np.load(tmp_path / f"r{k}.npy")) == (it, deff, conv)
        assert np.array_equal(np.load(tmp_path / f"mfl{k}.npy"), MFL)


def test_host_assembled_system_drop_in(pkg, oracle):
    """The reference's own arrays (A AoS, b, D) go in unchanged: set_system + solve."""
    rng = np.random.default_rng(11)
    nx, ny = 96, 64
    pix = rand_mask(rng, nx, ny)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    it, deff, conv, x, MFL, MFR = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-4, 40000, check_every=1000)
    for kernel in ("explicit", "scalar", "auto"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_system(A, b, D, 0.0, 1.0)
            s.set_field(x0)
            r = s.solve(1e-4, 40000, check_every=1000)
            # "auto": the rows of a piecewise-constant system are few, so the library harvests their
            # dictionary and runs the host-assembled system on the temporally blocked matrix-free kernel
            assert s.kernel_in_use() == ("matfree_tb" if kernel == "auto" else kernel)
            assert r.iters == it and r.deff_raw == deff and r.conv == conv
            assert_field(s.get_field(), x)
            assert np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)
            A2, b2 = s.get_system()
            assert np.array_equal(A2, A) and np.array_equal(b2, b)


def test_row_dictionary_limits_and_general_rhs(pkg, oracle):
    """A system whose rows are all different cannot be dictionary-coded and stays on the explicit
    kernel; one with a right-hand side away from the walls can, with b looked up everywhere."""
    rng = np.random.default_rng(21)
    nx, ny = 64, 40
    D = rng.uniform(0.5, 2.0, size=(ny, nx))            # every cell its own diffusivity
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want = oracle.sweeps(A, b, x0, 9)
    with pkg.Solver(nx, ny) as s:
        s.set_system(A, b, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(9)
        assert s.kernel_in_use() == "explicit"
        assert_field(s.get_field(), want)
    pix = rand_mask(rng, nx, ny)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    b = b.copy()
    b[::7] += 0.125                                     # sources inside the domain
    want = oracle.sweeps(A, b, x0, 9)
    for dict_on in (1, 0):
        with pkg.Solver(nx, ny) as s:
            s.set_tuning("dict", dict_on)
            s.set_system(A, b, D, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(9)
            assert s.kernel_in_use() == ("matfree_tb" if dict_on else "explicit")
            assert_field(s.get_field(), want)


def test_impermeable_solid_system(pkg, oracle, small_cases):
    """3-phase style rows (identity rows for Grid 1/2, Ds = 0 links) through the explicit kernel."""
    name = "s33x17"
    pix = small_cases[name + "_pix"]
    Ds, Df, CL, CR = small_cases[name + "_par"]
    ny, nx = pix.shape
    A, b = small_cases[name + "_Aimp"], small_cases[name + "_bimp"]
    x0 = oracle.linear_guess(nx, ny, CL, CR)
    with np.errstate(all="ignore"):
        want = oracle.sweeps(A, b, x0, 50)
    D0 = oracle.fill_D_2phase(pix, Df, 0.0)
    for kernel in ("explicit", "scalar", "auto", "matfree"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.assemble_from_D(D0, CL, CR, grid=small_cases[name + "_grid"])
            s.set_field(x0)
            s.sweeps(50)
            assert_field(s.get_field(), want)


# --------------------------------------------------------------- end to end

def test_img00000_config1_end_to_end(pkg, oracle, recorded, img00000):
    """Config #1 (00000.jpg, Ds 1e-3, tol 1e-6): same 110 001 sweeps, Deff within
    1e-8 of both recorded reference values, field within 1e-6 rel L2 of the golden field."""
    rec = recorded["img00000_2phase_batch"]
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "img00000_field.npy"))
    for kernel in KERNELS:
        with pkg.Solver(128, 128, kernel=kernel) as s:
            s.set_image(img00000)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            r = s.solve(1e-6, 500000)
            assert r.iters == rec["iters"]
            deff = r.deff_raw / 1.0                     # normalisation cuh:2017
            for key in ("deff_build_a", "deff_build_b"):
                assert abs(deff - rec[key]) / rec[key] <= DEFF_TOL
            assert deff == rec["deff_build_b"] or deff == rec["deff_build_a"]
            assert_field(s.get_field(), gold)
            assert r.checks == 12


@pytest.mark.parametrize("max_iter,check_every,tol", [(1, 10000, 1e-6), (5, 10000, 1e-6), (10001, 10000, 1e-6),
                                                      (10000, 10000, 1e-6), (25, 7, 1e-12), (8, 7, 1e-12),
                                                      (0, 10000, 1e-6), (300, 10, 1e-2)])
def test_stopping_rule_edges(pkg, oracle, max_iter, check_every, tol):
    rng = np.random.default_rng(5)
    nx, ny = 40, 24
    pix = rand_mask(rng, nx, ny)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-1)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    it, deff, conv, x, _, _ = oracle.jacobi(A, b, x0, D, 0.0, 1.0, tol, max_iter, check_every=check_every)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-1, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(tol, max_iter, check_every=check_every)
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        if max_iter > 0:
            assert_field(s.get_field(), x)


def test_two_phase_Ds0_nan_semantics(pkg, oracle, img00000):
    """Reference behaviour: Ds = 0 makes A0 = 0 cells, Deff = NaN, loop exits after 1 sweep."""
    for kernel in KERNELS:
        with pkg.Solver(128, 128, kernel=kernel) as s:
            s.set_image(img00000)
            s.assemble_2phase(0.0, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            r = s.solve(1e-6, 500000)
            assert r.iters == 1 and np.isnan(r.deff_raw)


def test_warm_start_continuation(pkg, oracle, img00000):
    """x is in/out (cuh:1163): a second solve starts from the first one's field,
    which is what the DCF/DCG continuation ramps rely on (cuh:1759-1817)."""
    D1 = oracle.fill_D_2phase(img00000, 100.0, 1e-3)
    A1, b1 = oracle.discretize(D1, 0.0, 1.0)
    x0 = oracle.linear_guess(128, 128, 0.0, 1.0)
    it1, d1, c1, x1, _, _ = oracle.jacobi(A1, b1, x0, D1, 0.0, 1.0, 1e-3, 30000)
    D2 = oracle.fill_D_2phase(img00000, 10000.0, 1e-3)
    A2, b2 = oracle.discretize(D2, 0.0, 1.0)
    it2, d2, c2, x2, _, _ = oracle.jacobi(A2, b2, x1, D2, 0.0, 1.0, 1e-3, 30000)
    with pkg.Solver(128, 128) as s:
        s.set_image(img00000)
        s.assemble_2phase(1e-3, 100.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r1 = s.solve(1e-3, 30000)
        s.assemble_2phase(1e-3, 10000.0, 0.0, 1.0)
        r2 = s.solve(1e-3, 30000)
        assert (r1.iters, r1.deff_raw) == (it1, d1)
        assert (r2.iters, r2.deff_raw, r2.conv) == (it2, d2, c2)
        assert_field(s.get_field(), x2)


# ------------------------------------------------ benchmark sizes (properties)

@pytest.mark.parametrize("n", [128, 1024, 4096])
def test_first_check_deff_at_benchmark_sizes(pkg, recorded, n):
    """Device generator + assembly + one sweep + flux at BASELINE sizes against the
    reference's recorded first-check Deff (bit-exact), for every kernel."""
    want = recorded["synthetic_first_check_deff"][str(n)]
    for kernel in KERNELS:
        with pkg.Solver(n, n, kernel=kernel) as s:
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            r = s.solve(1e-6, 1)
            assert r.iters == 1 and r.checks == 1
            assert abs(r.deff_raw - want) / want <= DEFF_TOL
            assert r.deff_raw == want, (kernel, n, repr(r.deff_raw))


def test_kernels_agree_at_4096(pkg):
    """Full benchmark size: the three kernels produce the same bits after 25 sweeps."""
    fields = []
    for kernel in KERNELS:
        with pkg.Solver(4096, 4096, kernel=kernel) as s:
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(25)
            fields.append(s.get_field())
    for f in fields[1:]:
        assert np.array_equal(fields[0], f)


def test_parallel_stripes_analytic_at_2048(pkg):
    """doc 5.3 eq. (7) at a large size: the linear guess is the exact solution of the
    parallel-stripe problem, so Deff = eps*Df + (1-eps)*Ds after the minimum 10 001 sweeps."""
    n, eps, Ds = 2048, 0.25, 1e-3
    pix = np.full((n, n), 255, dtype=np.uint8)
    pix[: int(eps * n), :] = 0
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_2phase(Ds, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-6, 500000)
        assert r.iters == 10001
        exact = eps + (1 - eps) * Ds
        assert abs(r.deff_raw - exact) / exact < 1e-11


def test_mirror_symmetry_at_1024(pkg):
    """Flipping the image top-bottom flips the field (N and S swap places in sigma,
    so agreement is to rounding, not bitwise)."""
    n = 1024
    with pkg.Solver(n, n) as s:
        s.synth_image(12345, 0)
        pix = s.get_image()
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(200)
        a = s.get_field()
        s.set_image(np.ascontiguousarray(pix[::-1]))
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(200)
        bflip = s.get_field()[::-1]
    assert rel_l2(bflip, a) < 1e-13


def test_reference_second_image_shape_1002x2007(pkg, oracle):
    """The shape of the reference's other shipped image (00042.jpg, 1002 x 2007): non-power-of-two
    strips and chunks on every kernel, odd row count, 2 and 3 phases."""
    nx, ny = 1002, 2007
    rng = np.random.default_rng(42)
    f = np.kron(rng.random((ny // 9 + 1, nx // 6 + 1)), np.ones((9, 6)))[:ny, :nx]
    pix = np.where(f < 0.5, 0, np.where(f < 0.72, 150, 255)).astype(np.uint8)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    want = oracle.sweeps(A, b, x0, 21)
    for kernel in KERNELS:
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_image(pix)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(21)
            assert_field(s.get_field(), want)
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D3 = oracle.fill_D_3phase(pix, 1.0, 0.0, 1237500.0)
    with np.errstate(all="ignore"):
        A3, b3 = oracle.discretize(D3, 0.0, 1.0, grid=grid)
        want3 = oracle.sweeps(A3, b3, x0, 21)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 1237500.0, 0.0, 1.0, grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(21)
        assert s.kernel_in_use() == "matfree_tb"
        assert_field(s.get_field(), want3)


@pytest.mark.parametrize("Ds,Df,CL,CR", [(1e-6, 1.0, 0.0, 1.0), (1.0, 1237500.0, 0.0, 1.0), (0.3, 2.0, 2.0, -1.0),
                                         (5e-324, 1.0, 0.0, 1.0)])
def test_extreme_contrasts_and_boundary_values(pkg, oracle, Ds, Df, CL, CR):
    """Phase contrasts of 1e6 (the range the reference's documentation validates, doc 5.3), a gas-like
    1 237 500, reversed / negative wall concentrations, a denormal diffusivity: same bits as the oracle."""
    nx, ny = 100, 100
    pix = np.full((ny, nx), 255, dtype=np.uint8)
    pix[:, :37] = 0                              # series stripes
    pix[40:60, 50:80] = 0
    D = oracle.fill_D_2phase(pix, Df, Ds)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, CL, CR)
        x0 = oracle.linear_guess(nx, ny, CL, CR)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, x0, D, CL, CR, 1e-5, 3000, check_every=500)
    for kernel in ("matfree_tb", "explicit"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_image(pix)
            s.assemble_2phase(Ds, Df, CL, CR)
            s.init_linear(CL, CR)
            r = s.solve(1e-5, 3000, check_every=500)
            assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv) or (np.isnan(deff) and np.isnan(r.deff_raw))
            assert_field(s.get_field(), x)


def test_abi_misuse_returns_codes_not_crashes(pkg):
    """Raw C ABI with NULL pointers, bad sizes and wrong call order: a negative code and a message
    every time, never a crash, and the context stays usable."""
    import ctypes as C
    from effectivediffusivityfvm_amd import _capi
    L = _capi.load()
    ctx = C.c_void_p()
    bad = []

    def expect_fail(rc, what):
        if rc >= 0 or not L.deff_last_error():
            bad.append(what)

    expect_fail(L.deff_create(0, 1, 8, C.byref(ctx)), "nx = 1")
    expect_fail(L.deff_create(0, 8, 8, None), "out = NULL")
    expect_fail(L.deff_create(99, 8, 8, C.byref(ctx)), "device 99")
    expect_fail(L.deff_create_batch(0, 8, 8, 0, C.byref(ctx)), "nimg = 0")
    expect_fail(L.deff_create_batch(0, 65536, 65536, 1, C.byref(ctx)), "2^32 cells")
    assert L.deff_create(0, 24, 16, C.byref(ctx)) == 0
    res = _capi.Result()
    expect_fail(L.deff_solve(ctx, 2 / 3, 1e-3, 10, 10, C.byref(res), None, None), "solve before anything")
    expect_fail(L.deff_assemble_2phase(ctx, 1e-3, 1.0, 0.0, 1.0), "assemble without image")
    expect_fail(L.deff_set_tuning(ctx, b"no_such_key", 1), "unknown tuning key")
    expect_fail(L.deff_set_tuning(ctx, b"tb_T", -1), "negative tuning value")
    expect_fail(L.deff_set_kernel(ctx, 77), "unknown kernel id")
    v = C.c_int()
    expect_fail(L.deff_get_plan(ctx, b"nope", C.byref(v)), "unknown plan key")
    pix = np.zeros((16, 24), dtype=np.uint8)
    expect_fail(L.deff_set_image(ctx, pix, 24, 16, 2, 1), "amplification does not match the mesh")
    expect_fail(L.deff_set_image(ctx, pix, 0, 16, 1, 1), "W = 0")
    assert L.deff_set_image(ctx, pix, 24, 16, 1, 1) == 0
    assert L.deff_assemble_2phase(ctx, 1e-3, 1.0, 0.0, 1.0) == 0
    expect_fail(L.deff_sweeps(ctx, 3, 2 / 3, None), "sweeps without a field")
    assert L.deff_init_linear(ctx, 0.0, 1.0) == 0
    expect_fail(L.deff_sweeps(ctx, -1, 2 / 3, None), "negative sweep count")
    expect_fail(L.deff_solve(ctx, 2 / 3, 1e-3, 10, 0, C.byref(res), None, None), "check_every = 0")
    expect_fail(L.deff_solve(ctx, 2 / 3, 1e-3, 10, 10, None, None, None), "result = NULL")
    grp = C.c_void_p()
    expect_fail(L.deff_slab_group_create(4, None, 64, 20, C.byref(grp)), "5 rows per slab")
    expect_fail(L.deff_slab_group_create(0, None, 64, 64, C.byref(grp)), "0 slabs")
    expect_fail(L.deff_slab_rank_create(0, 64, 64, 3, 2, b"x" * 128, C.byref(grp)), "rank >= nranks")
    # and after all that the context still works
    assert L.deff_solve(ctx, 2 / 3, 1e-3, 50, 10, C.byref(res), None, None) == 0 and res.iters > 0
    assert L.deff_destroy(ctx) == 0
    assert L.deff_destroy(None) == 0
    assert not bad, bad


def test_errors_are_loud(pkg):
    with pkg.Solver(16, 16) as s:
        with pytest.raises(pkg.DeffError):
            s.sweeps(1)                      # no field / system
        s.init_linear(0.0, 1.0)
        with pytest.raises(pkg.DeffError):
            s.sweeps(1)                      # no system
        with pytest.raises(pkg.DeffError):
            s.set_image(np.zeros((8, 8), dtype=np.uint8))   # shape mismatch


def test_wall_column_links_keep_linear_addressing(pkg, oracle):
    """A caller-built A may link a wall column to the neighbouring ROW (A[p][1] != 0 in column 0 reads x[p-1], the
    previous row's last cell: the reference's kernel addresses linearly, cuh:80-83).  The reference's own assembly
    never produces such links (cuh:849-864); when a caller does, the system must stay on the explicit kernels, which
    keep that addressing, and match the oracle -- not be moved to the dictionary kernels, whose walls have no
    outer neighbour.  An odd width (padded rows) cannot represent it and is refused."""
    rng = np.random.default_rng(5)
    nx, ny = 64, 24
    pix = rand_mask(rng, nx, ny)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    A = A.copy().reshape(ny, nx, 5)
    A[1:, 0, 1] = -0.125            # W link of column 0 -> previous row's last cell (row 0 would read x[-1]: left alone)
    A[:-1, -1, 2] = -0.25           # E link of the last column -> next row's first cell
    A = A.reshape(-1, 5)
    x0 = rng.random((ny, nx))
    want = oracle.sweeps(A, b, x0, 11)
    for kernel in ("auto", "matfree_tb", "explicit", "scalar"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_system(A, b, D, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(11)
            assert s.kernel_in_use() in ("explicit", "scalar")
            assert_field(s.get_field(), want)
    # the same rows without the wrap links ARE dictionary-coded (the control)
    A2, b2 = oracle.discretize(D, 0.0, 1.0)
    with pkg.Solver(nx, ny) as s:
        s.set_system(A2, b2, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(11)
        assert s.kernel_in_use() == "matfree_tb"
        assert_field(s.get_field(), oracle.sweeps(A2, b2, x0, 11))
    # odd width: refused, with a message
    nxo = 63
    Do = oracle.fill_D_2phase(rand_mask(rng, nxo, ny), 1.0, 1e-2)
    Ao, bo = oracle.discretize(Do, 0.0, 1.0)
    Ao = Ao.copy().reshape(ny, nxo, 5)
    Ao[3, 0, 1] = -0.5
    with pkg.Solver(nxo, ny) as s:
        with pytest.raises(pkg.DeffError, match="wall column"):
            s.set_system(Ao.reshape(-1, 5), bo, Do, 0.0, 1.0)


def test_stream_callback_errors_surface(pkg, monkeypatch):
    """An exception inside the image-done callback of solve_stream must be re-raised after the call (ctypes would
    otherwise swallow it and the caller would see a KeyError much later)."""
    import effectivediffusivityfvm_amd.solver as solver_mod

    class Boom(RuntimeError):
        pass

    def broken():
        raise Boom("result object could not be built")

    rng = np.random.default_rng(2)
    imgs = [rand_mask(rng, 32, 16) for _ in range(3)]
    with pkg.Solver(32, 16, nimg=2) as s:
        monkeypatch.setattr(solver_mod, "SolveResult", broken)
        with pytest.raises(Boom):
            s.solve_stream(iter(imgs), 1e-2, 1.0, 0.0, 1.0, 1e-3, 200, check_every=50)


@pytest.mark.parametrize("B", [1, 3])
def test_device_side_flux_sums(pkg, oracle, B):
    """flux_reduce = 1: the wall fluxes are added up on the device IN ROW ORDER (the reference's order, cuh:1258-1259), one
    wave per image, so a check moves 16 bytes per image instead of 16 per row: iteration counts, Deff and conv must be the
    bits of the host-summed run and of the oracle.  flux_reduce = 2: a wave-level tree (fixed order, not the reference's):
    Deff within 1e-13, inside the north-star's 1e-8, and the same stopping decisions on this case."""
    nx, ny = 130, 70
    rng = np.random.default_rng(17 + B)
    pixs = [rand_mask(rng, nx, ny, 0.4 + 0.1 * k) for k in range(B)]
    want = []
    for k in range(B):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        want.append(oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-4, 6000, check_every=200))
    for mode in (0, 1, 2):
        for fluxes in (True, False):
            with pkg.Solver(nx, ny, nimg=B) as s:
                s.set_tuning("flux_reduce", mode)
                s.set_image(np.stack(pixs) if B > 1 else pixs[0])
                s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
                s.init_linear(0.0, 1.0)
                res = s.solve(1e-4, 6000, check_every=200, fluxes=fluxes)
                got = s.get_field()
                d_after, _, _ = s.flux()
            res = res if isinstance(res, list) else [res]
            for k in range(B):
                it, deff, conv, x, MFL, MFR = want[k]
                assert res[k].iters == it
                if mode < 2:
                    assert (res[k].deff_raw, res[k].conv) == (deff, conv)
                else:
                    assert abs(res[k].deff_raw - deff) <= 1e-13 * abs(deff)
                if fluxes:
                    assert np.array_equal(res[k].MFL, MFL) and np.array_equal(res[k].MFR, MFR)
                assert_field(got[k * ny:(k + 1) * ny], x)


def test_dropped_kernel_forms_are_not_tuning_values(pkg):
    """The paired-wave streaming form of round 3 (tb_impl = 3: measured 6-7 % slower) is no longer part of
    the library; the key takes 0, 1 or 2."""
    with pkg.Solver(64, 64) as s:
        for v in (0, 1, 2):
            s.set_tuning("tb_impl", v)
        with pytest.raises(pkg.DeffError, match="tb_impl takes"):
            s.set_tuning("tb_impl", 3)
