"""The workgroup-tile form of the temporally blocked pass (csrc/kernels_wgtile.hpp: 8 waves share a tile that stays in
registers for the T sweeps of a pass, first/last rows through an LDS mailbox, one barrier per sweep) against the CPU
oracle: every raggedness of strips / row tiles, stacks with frozen images, row slabs, the zero-diffusivity guard, the
contracted arithmetic.  The form is selected with the tuning knob tb_impl = 2 (1 = streaming kernel); results must be
bit-identical to the oracle whatever the form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


def rand_mask(rng, nx, ny, p=0.5):
    return np.where(rng.random((ny, nx)) < p, 0, 255).astype(np.uint8)


def assert_field(got, want):
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got))
    assert np.linalg.norm(got[fin] - want[fin]) <= 1e-6 * np.linalg.norm(want[fin])
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("T", [4, 8])
@pytest.mark.parametrize("R,LY", [(4, 0), (6, 0), (7, 0), (6, 11), (7, 5)])
@pytest.mark.parametrize("shape", [(600, 300), (1030, 37), (130, 70), (256, 256), (122, 9), (2, 64), (498, 40), (250, 33),
                                   (97, 41), (1001, 333)])
def test_wgtile_sweeps_vs_oracle(pkg, oracle, shape, T, R, LY):
    nx, ny = shape
    rng = np.random.default_rng(nx * 7 + ny * 13 + T + R)
    pix = rand_mask(rng, nx, ny, 0.55)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    nsw = 3 * T + 3
    want = oracle.sweeps(A, b, x0, nsw)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_T", T)
        s.set_tuning("tb_impl", 2)
        s.set_tuning("tb_R", R)
        s.set_tuning("tb_LY", LY)
        s.set_tuning("tb_wall_halo", (nx + ny + T) % 3)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(nsw)
        p = s.plan()
        if ny >= 8:
            assert s.kernel_in_use() == "matfree_tb" and p["tb_impl"] == 2 and p["tb_R"] == R and p["tb_T"] == T
            assert 1 <= p["tb_LY"] <= 8 * R - 2 * T and p["tb_LY"] * p["tb_chunks_per_image"] >= ny
            if LY:
                assert p["tb_LY"] <= LY
        assert_field(s.get_field(), want)


@pytest.mark.parametrize("T,R", [(4, 7), (8, 4), (8, 6)])
def test_wgtile_omega_one_boundary_values_and_fma(pkg, oracle, T, R):
    """updateX_V1's arithmetic (omega = 1), non-trivial wall values, and the contracted arithmetic against the
    oracle's -ffp-contract=fast build."""
    nx, ny = 300, 150
    rng = np.random.default_rng(T * 10 + R)
    pix = rand_mask(rng, nx, ny, 0.45)
    D = oracle.fill_D_2phase(pix, 2.0, 0.3)
    A, b = oracle.discretize(D, 2.0, -1.0)
    x0 = rng.random((ny, nx))
    for omega, kern in ((1.0, 1), (2.0 / 3.0, 0)):
        for flavour, fma in ((None, 0), ("fma", 1)):
            want = oracle.sweeps(A, b, x0, 2 * T + 1, kernel=kern, omega=omega, flavour=flavour)
            with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
                s.set_tuning("tb_T", T); s.set_tuning("tb_impl", 2); s.set_tuning("tb_R", R); s.set_tuning("fma", fma)
                s.set_image(pix)
                s.assemble_2phase(0.3, 2.0, 2.0, -1.0)
                s.set_field(x0)
                s.sweeps(2 * T + 1, omega)
                assert s.plan()["tb_impl"] == 2
                assert_field(s.get_field(), want)


def test_wgtile_zero_diffusivity_guard_and_three_phase(pkg, oracle, img00000):
    """Ds = 0: -0.0 links and NaN cells (the reference's non-zero test, cuh:77) through the guarded instantiation; then
    the as-shipped 3-phase system (harvested dictionary, right-hand side possibly away from the walls)."""
    D = oracle.fill_D_2phase(img00000, 1.0, 0.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0)
        want = oracle.sweeps(A, b, oracle.linear_guess(128, 128, 0.0, 1.0), 19)
    with pkg.Solver(128, 128, kernel="matfree_tb") as s:
        s.set_tuning("tb_T", 8); s.set_tuning("tb_impl", 2)
        s.set_image(img00000)
        s.assemble_2phase(0.0, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(19)
        assert s.plan()["tb_impl"] == 2
        assert_field(s.get_field(), want)
    grid, _ = oracle.floodfill((img00000 > 200).astype(np.uint32))
    D3 = oracle.fill_D_3phase(img00000, 1.0, 0.0, 1237500.0)
    with np.errstate(all="ignore"):
        A3, b3 = oracle.discretize(D3, 0.0, 1.0, grid=grid)
        want3 = oracle.sweeps(A3, b3, oracle.linear_guess(128, 128, 0.0, 1.0), 21)
    for R in (4, 7):
        with pkg.Solver(128, 128) as s:
            s.set_tuning("tb_T", 4); s.set_tuning("tb_impl", 2); s.set_tuning("tb_R", R)
            s.set_image(img00000)
            s.assemble_3phase(0.0, 1.0, 1237500.0, 0.0, 1.0, grid)
            s.init_linear(0.0, 1.0)
            s.sweeps(21)
            assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_impl"] == 2
            assert_field(s.get_field(), want3)


@pytest.mark.parametrize("shape,B", [((130, 70), 5), ((64, 9), 3), ((256, 128), 2)])
def test_wgtile_stack_with_frozen_images(pkg, oracle, shape, B):
    """Stacks: tiles never straddle images; a solve in which images stop at different checks (device mask)."""
    nx, ny = shape
    rng = np.random.default_rng(nx + B)
    pixs = [rand_mask(rng, nx, ny, 0.35 + 0.1 * k) for k in range(B)]
    with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
        s.set_tuning("tb_T", 4); s.set_tuning("tb_impl", 2); s.set_tuning("tb_R", 4)
        s.set_image(np.stack(pixs))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-3, 4000, check_every=200)
        got = s.get_field()
        assert s.plan()["tb_impl"] == 2
    res = res if isinstance(res, list) else [res]
    iters = set()
    for k in range(B):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 4000,
                                                check_every=200)
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
        assert_field(got[k * ny:(k + 1) * ny], x)
        iters.add(it)
    if B > 2:
        assert len(iters) > 1                      # the device mask was exercised


@pytest.mark.parametrize("nslabs", [2, 3])
def test_wgtile_row_slabs(pkg, oracle, nslabs):
    """Row slabs (halo rows above and below the owned rows, mesh rows taken from the whole image)."""
    nx, NY = 384, 203
    rng = np.random.default_rng(nslabs)
    pix = rand_mask(rng, nx, NY, 0.55)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((NY, nx))
    want = oracle.sweeps(A, b, x0, 29)
    with pkg.SlabGroup(nx, NY, [0] * nslabs) as g:
        g.set_tuning("tb_T", 8); g.set_tuning("tb_impl", 2); g.set_tuning("tb_R", 4)
        g.set_image(pix)
        g.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        g.set_field(x0)
        g.sweeps(29)
        assert all(p["tb_impl"] == 2 and p["tb_T"] == 8 for p in g.plans())
        assert_field(g.get_field(), want)
        d, MFL, MFR = g.flux()
        dor, MFLo, MFRo = oracle.flux_deff(want, D, 0.0, 1.0)
        assert d == dor and np.array_equal(MFL, MFLo) and np.array_equal(MFR, MFRo)


def test_wgtile_1024_vs_oracle(pkg, oracle):
    """BASELINE config #2's image: 1024^2 synthetic, 27 sweeps, both forms, against the oracle."""
    n = 1024
    pix = oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    want = oracle.sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 27)
    for impl, T, R in ((2, 8, 7), (2, 8, 6), (2, 4, 4), (1, 4, 0), (0, 0, 0)):
        with pkg.Solver(n, n, kernel="matfree_tb") as s:
            s.set_tuning("tb_T", T); s.set_tuning("tb_impl", impl); s.set_tuning("tb_R", R)
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(27)
            assert_field(s.get_field(), want)


@pytest.mark.parametrize("launch", [0, 1])               # tuning key tb_launch
@pytest.mark.parametrize("T,R", [(8, 7), (8, 4), (4, 6), (8, 6)])
@pytest.mark.parametrize("shape", [(600, 300), (1030, 37), (130, 70), (256, 256), (2, 64), (1001, 333), (1024, 1024)])
def test_resident_passes_vs_oracle(pkg, oracle, shape, T, R, launch):
    """Resident passes (k_sweep_wgres): all tiles on the chip, the passes of a batch in ONE launch, neighbours synchronised
    through per-tile flags (tb_launch = 0, the default), and switched off (1: one launch per pass).
    5 passes + 3 single sweeps, then 2 more passes (a second launch: epoch counters carry over), against the oracle."""
    nx, ny = shape
    if (nx, ny) == (1024, 1024) and (T, R) != (8, 7):
        pytest.skip("1024^2 once, on the planner's own tile")
    rng = np.random.default_rng(nx * 3 + ny * 5 + T + R)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    n1, n2 = 5 * T + 3, 2 * T
    want1 = oracle.sweeps(A, b, x0, n1)
    want2 = oracle.sweeps(A, b, want1, n2)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_T", T); s.set_tuning("tb_impl", 2); s.set_tuning("tb_R", R)
        s.set_tuning("tb_launch", launch)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(n1)
        p = s.plan()
        assert p["tb_impl"] == 2 and p["tb_T"] == T
        tiles = p["tb_strips"] * p["tb_chunks_per_image"]
        expect = launch != 1 and tiles <= 256
        assert p["tb_resident"] == int(expect), p
        launches, _ = s.last_launches()
        assert launches == (1 + 3 if expect else 5 + 3)
        assert_field(s.get_field(), want1)
        s.sweeps(n2)
        assert_field(s.get_field(), want2)


def test_resident_stack_with_frozen_images(pkg, oracle):
    """A stack of images on resident passes: tiles wait only for tiles of their own image, and a frozen image's tiles
    take no part (solve_batch freezes each image at ITS stopping rule); per-image iteration counts, Deff and fields
    must equal one-image-at-a-time solves on the per-pass path."""
    nx = ny = 128
    B = 6
    rng = np.random.default_rng(77)
    imgs = [rand_mask(rng, nx, ny, 0.35 + 0.05 * k) for k in range(B)]
    res = {}
    for resident in (1, 0):
        with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_launch", 0 if resident else 1)
            s.set_image(np.stack(imgs))
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            r = s.solve(1e-3, 200000, check_every=500)
            assert s.plan()["tb_resident"] == resident
            res[resident] = (r, s.get_field())
    ra, fa = res[1]
    rb, fb = res[0]
    assert [x.iters for x in ra] == [x.iters for x in rb] and len({x.iters for x in ra}) > 1
    assert [x.deff_raw for x in ra] == [x.deff_raw for x in rb]
    assert np.array_equal(fa, fb)


def test_resident_launch_that_gives_up_is_redone(pkg, oracle):
    """The failure path of resident passes: a tile that never publishes (test hook tb_debug_stall) leaves its neighbours
    polling; after the bounded wait (2 s) one of them raises the abort flag and every workgroup returns.  The solve is NOT
    lost: the library restores the field the interval started from, redoes the interval with one launch per pass and
    keeps launching that way -- same iteration count, Deff and field as the oracle, the fallback reported through
    deff_get_plan("tb_fallbacks").  Through deff_sweeps() and through the solve loop (several checks, a stall in the
    second interval)."""
    import time
    nx = ny = 512
    pix = oracle.synth_mask(nx, ny, 7, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    want64 = oracle.sweeps(A, b, x0, 64)
    want131 = oracle.sweeps(A, b, want64, 67)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(64)
        assert s.plan()["tb_resident"] == 1 and s.plan_value("tb_fallbacks") == 0
        assert np.array_equal(s.get_field(), want64)
        s.set_tuning("tb_debug_stall", 12)
        t0 = time.perf_counter()
        s.sweeps(67)                                       # 8 resident passes that abort + 3 single sweeps: all redone
        assert 1.5 < time.perf_counter() - t0 < 10.0
        assert s.plan_value("tb_fallbacks") == 1
        assert np.array_equal(s.get_field(), want131)
        s.sweeps(0)
        assert s.plan()["tb_resident"] == 0                # the context stays on one launch per pass
    it, deff, conv, want, _, _ = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-9, 1201, check_every=400)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        seen = []

        def on_check(k, d, ch):
            seen.append(k)
            if k == 400:
                s.set_tuning("tb_debug_stall", 12)         # the interval after the second check stalls
        s.set_progress(on_check)
        r = s.solve(1e-9, 1201, check_every=400)
        assert seen == [0, 400, 800, 1200]
        assert s.plan_value("tb_fallbacks") == 1
        assert r.iters == it and r.deff_raw == deff and r.conv == conv
        assert np.array_equal(s.get_field(), want)


_TWO_PROC_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import effectivediffusivityfvm_amd as pkg
with pkg.Solver(1024, 1024, kernel="matfree_tb") as s:
    s.synth_image(12345, 0)
    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
    s.init_linear(0.0, 1.0)
    r = s.solve(1e-12, 150001)
    print(json.dumps(dict(iters=r.iters, deff=r.deff_raw, conv=r.conv, resident=s.plan()["tb_resident"],
                          fallbacks=s.plan_value("tb_fallbacks"), sha=hashlib.sha256(s.get_field().tobytes()).hexdigest())))
"""


def test_two_processes_resident_on_one_gpu(pkg, tmp_path):
    """What one process cannot rule out: ANOTHER PROCESS's resident launch on the same GPU.  Two fresh child processes solve
    the same 1024^2 image (150 001 sweeps, 16 checks) side by side; their resident grids may each hold part of the chip, in
    which case bounded waits run out and the library redoes the interval with one launch per pass.  Either way both must
    finish, with the same sweeps / Deff / conv / field as a child that has the GPU to itself.  Run once."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "child.py"
    script.write_text(_TWO_PROC_CHILD)

    def parse(out):
        return json.loads(out.strip().splitlines()[-1])
    alone = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert alone.returncode == 0, alone.stderr
    ref = parse(alone.stdout)
    assert ref["iters"] == 150001 and ref["fallbacks"] == 0 and ref["resident"] == 1
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(2)]
    try:
        outs = [p.communicate(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
        got = parse(o)
        assert (got["iters"], got["deff"], got["conv"], got["sha"]) == (ref["iters"], ref["deff"], ref["conv"], ref["sha"]), got
    print("two processes: fallbacks", [parse(o)["fallbacks"] for o, _ in outs])


def test_resident_passes_with_a_second_context_loading_the_gpu(pkg):
    """Resident launches share the GPU with another context's kernels (a host thread sweeping a 4096^2 image on its own
    stream): 3 x 80 000 sweeps of a 1024^2 image = 30 000 passes x 234 tiles of flag-synchronised halo exchange must give the
    same field, bit for bit, as one launch per pass on a quiet GPU -- the rim stores are acknowledged before a tile's flag
    is raised whatever else loads the memory system -- and must not run into the bounded wait."""
    import hashlib
    import threading
    n, nsw = 1024, 80_000

    def solve(launch):
        with pkg.Solver(n, n, kernel="matfree_tb") as s:
            s.set_tuning("tb_launch", launch)
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(nsw)
            assert s.plan()["tb_resident"] == (0 if launch == 1 else 1)
            return hashlib.sha256(s.get_field().tobytes()).hexdigest()

    ref = solve(1)
    stop = threading.Event()
    launches = []

    def hammer():
        with pkg.Solver(4096, 4096, kernel="matfree_tb") as h:
            h.synth_image(1, 0)
            h.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            h.init_linear(0.0, 1.0)
            k = 0
            while not stop.is_set():
                h.sweeps(40)
                k += 1
            launches.append(k)

    th = threading.Thread(target=hammer)
    th.start()
    try:
        got = [solve(0) for _ in range(3)]
    finally:
        stop.set()
        th.join()
    assert launches and launches[0] >= 3
    assert got == [ref] * 3


@pytest.mark.parametrize("n,want_nw", [(1024, 12), (640, 8), (1152, 12), (1200, 12), (1536, 16), (2048, 16)])
def test_long_solve_resident_equals_one_launch_per_pass(pkg, n, want_nw):
    """short parity tests do not catch exchange races (round 3's mailbox variant passed all of them and got
    one 150 001-sweep solve wrong), so the suite itself holds long ones: 150 001 sweeps through the solve loop (16 checks,
    ~18 750 flag-synchronised passes per tile) on each resident form -- 12-wave link-symmetric tiles (1024^2; 1152^2 with passes
    of six sweeps, 1200^2 with passes of four), 8-wave tiles (640^2), tall 16-wave tiles (1536^2, 2048^2) -- must give the SHA-256 of the field, the Deff and the last change of one
    launch per pass."""
    import hashlib

    def solve(launch):
        with pkg.Solver(n, n, kernel="matfree_tb") as s:
            s.set_tuning("tb_launch", launch)
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            r = s.solve(1e-30, 150001)
            p = s.plan()
            assert p["tb_resident"] == (0 if launch == 1 else 1) and s.plan_value("tb_fallbacks") == 0
            if launch == 0:
                assert p["tb_NW"] == want_nw, p
            return r.iters, r.checks, r.deff_raw, r.conv, hashlib.sha256(s.get_field().tobytes()).hexdigest()

    a, b = solve(0), solve(1)
    assert a[0] == 150001 and a[1] == 16
    assert a == b


# ---- tall resident tiles: 16 waves x R rows, field in registers, codes in LDS, matrix rows looked up in every sweep ----

TALL_SHAPES = [(300, 200), (1030, 137), (600, 700), (250, 333), (2, 164), (97, 241), (1281, 410), (122, 9)]


@pytest.mark.parametrize("R", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("shape", TALL_SHAPES)
def test_tall_tiles_vs_oracle(pkg, oracle, shape, R):
    """k_sweep_wgres<.., TALL>: every R, ragged strips and row tiles, images shorter than one tile, one-cell-wide
    columns; 5 passes + 3 single sweeps, then 2 more passes in a second launch."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 11 + ny * 3 + R)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want1 = oracle.sweeps(A, b, x0, 43)
    want2 = oracle.sweeps(A, b, want1, 16)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", R)
        s.set_tuning("tb_wall_halo", (nx + ny + R) % 3)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(43)
        p = s.plan()
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"]) == (2, 16, R, 8, 1), p
        cpi = p["tb_chunks_per_image"]                        # no halo rows beyond a wall: 1 tile holds 16R rows, 2 tiles 16R - 8 each
        assert 1 <= p["tb_LY"] <= 16 * R - 8 * min(cpi - 1, 2) and p["tb_LY"] * cpi >= ny
        launches, _ = s.last_launches()
        assert launches == 1 + 3
        assert_field(s.get_field(), want1)
        s.sweeps(16)
        assert_field(s.get_field(), want2)


@pytest.mark.parametrize("R", [4, 6, 12, 14])
def test_tall_tiles_omega_one_boundary_values_fma_and_launch_modes(pkg, oracle, R):
    nx, ny = 300, 450
    rng = np.random.default_rng(R)
    pix = rand_mask(rng, nx, ny, 0.45)
    D = oracle.fill_D_2phase(pix, 2.0, 0.3)
    A, b = oracle.discretize(D, 2.0, -1.0)
    x0 = rng.random((ny, nx))
    for omega, kern in ((1.0, 1), (2.0 / 3.0, 0)):
        for flavour, fma in ((None, 0), ("fma", 1)):
            want = oracle.sweeps(A, b, x0, 25, kernel=kern, omega=omega, flavour=flavour)
            for launch in (0,):
                with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
                    s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", R); s.set_tuning("fma", fma)
                    s.set_tuning("tb_launch", launch)
                    s.set_image(pix)
                    s.assemble_2phase(0.3, 2.0, 2.0, -1.0)
                    s.set_field(x0)
                    s.sweeps(25, omega)
                    assert s.plan()["tb_NW"] == 16
                    assert np.array_equal(s.get_field(), want)
    # one launch per pass requested: tall tiles (which only exist resident) are not planned
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", R); s.set_tuning("tb_launch", 1)
        s.set_image(pix)
        s.assemble_2phase(0.3, 2.0, 2.0, -1.0)
        s.set_field(x0)
        s.sweeps(25)
        p = s.plan()
        assert p["tb_NW"] == 8 and p["tb_resident"] == 0
        assert np.array_equal(s.get_field(), oracle.sweeps(A, b, x0, 25))


def test_tall_tiles_zero_diffusivity_guard_and_three_phase(pkg, oracle, img00000):
    """The guarded instantiation (Ds = 0: -0.0 links, NaN cells) and a harvested dictionary (3 phases) on tall tiles."""
    pix = np.tile(img00000, (3, 2))                              # 256 x 384: several row tiles at R = 6
    ny, nx = pix.shape
    D = oracle.fill_D_2phase(pix, 1.0, 0.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0)
        want = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 19)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 6)
        s.set_image(pix)
        s.assemble_2phase(0.0, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(19)
        assert s.plan()["tb_NW"] == 16
        assert_field(s.get_field(), want)
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D3 = oracle.fill_D_3phase(pix, 1.0, 0.0, 1237500.0)
    with np.errstate(all="ignore"):
        A3, b3 = oracle.discretize(D3, 0.0, 1.0, grid=grid)
        want3 = oracle.sweeps(A3, b3, oracle.linear_guess(nx, ny, 0.0, 1.0), 21)
    with pkg.Solver(nx, ny) as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 8)
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 1237500.0, 0.0, 1.0, grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(21)
        assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_NW"] == 16
        assert_field(s.get_field(), want3)


@pytest.mark.parametrize("R", [5, 6, 7, 8, 9, 10, 11, 12])
def test_tall_tiles_with_rows_dealt_by_age(pkg, oracle, R):
    """k_sweep_wgage (round 4): the 4R rows of a SIMD's four waves dealt by the waves' age -- each age runs the pass loop
    instantiated for its own row count, the tile and its halo stay what they were.  Against the oracle and against equal rows,
    both arithmetics, ragged shapes (tiles cut by the image's bottom edge, one strip narrower than the rest), two launches."""
    for nx, ny in ((300, 16 * R + 40), (1030, 137), (250, 3 * (16 * R - 16) + 5)):
        rng = np.random.default_rng(nx + ny + R)
        pix = rand_mask(rng, nx, ny, 0.5)
        D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
        A, b = oracle.discretize(D, 0.0, 1.0)
        x0 = rng.random((ny, nx))
        for fma, flavour in ((0, None), (1, "fma")):
            want1 = oracle.sweeps(A, b, x0, 43, flavour=flavour)
            want2 = oracle.sweeps(A, b, want1, 16, flavour=flavour)
            for deal in (1, 0):
                with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
                    s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", R); s.set_tuning("fma", fma)
                    s.set_tuning("tb_tall_deal", deal)
                    s.set_image(pix)
                    s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
                    s.set_field(x0)
                    s.sweeps(43)
                    p = s.plan()
                    assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_resident"], p["tb_aged"]) == (2, 16, R, 1, deal), p
                    assert np.array_equal(s.get_field(), want1), (nx, ny, fma, deal)
                    s.sweeps(16)
                    assert np.array_equal(s.get_field(), want2), (nx, ny, fma, deal)


def test_tall_tiles_with_rows_dealt_by_age_on_a_system_that_is_not_link_symmetric(pkg, oracle):
    """The 3-phase assembly with impermeable solid (identity rows carry +0 links, their neighbours -0: not bit-symmetric, ten
    lookups per row) on the reference's own 00042.jpg, cropped: k_sweep_wgage's SYM = false variant against the oracle."""
    import os
    from conftest import ROOT
    full = pkg.load_jpeg_gray(os.path.join(ROOT, "tests", "golden", "00042.jpg"))
    pix = np.ascontiguousarray(full[:700, :600])
    ny, nx = pix.shape
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 1237500.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
        want = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 37)
    for deal in (1, 0):
        with pkg.Solver(nx, ny) as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 6); s.set_tuning("tb_tall_deal", deal)
            s.set_image(pix)
            s.assemble_3phase(0.0, 1.0, 1237500.0, 0.0, 1.0, grid)
            s.init_linear(0.0, 1.0)
            s.sweeps(37)
            p = s.plan()
            assert s.kernel_in_use() == "matfree_tb"
            assert (p["tb_NW"], p["tb_R"], p["tb_resident"], p["tb_sym"], p["tb_aged"]) == (16, 6, 1, 2, deal), p
            assert np.array_equal(s.get_field(), want, equal_nan=True)


def test_tall_tiles_stack_with_frozen_images(pkg, oracle):
    """A stack on tall tiles: images stop at different checks (their tiles then leave the launch at once)."""
    nx, ny, B = 130, 170, 5
    rng = np.random.default_rng(99)
    pixs = [rand_mask(rng, nx, ny, 0.35 + 0.1 * k) for k in range(B)]
    with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 6)
        s.set_image(np.stack(pixs))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-3, 6000, check_every=200)
        got = s.get_field()
        assert s.plan()["tb_NW"] == 16 and s.plan()["tb_resident"] == 1
    iters = set()
    for k in range(B):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 6000,
                                                check_every=200)
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
        assert_field(got[k * ny:(k + 1) * ny], x)
        iters.add(it)
    assert len(iters) > 1


def test_2048_default_plan_is_tall_and_matches_the_oracle(pkg, oracle):
    """2048^2 (4 Mi cells, where the streaming form used to take over): the planner's own choice -- 247 tiles of 176 rows --
    against the oracle, 27 sweeps, wall fluxes and Deff too; and what it replaces gives the same bits."""
    n = 2048
    pix = oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    want = oracle.sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 27)
    dor, MFLo, MFRo = oracle.flux_deff(want, D, 0.0, 1.0)
    del A, b
    for nw in (0, 8):
        with pkg.Solver(n, n) as s:
            s.set_tuning("tb_NW", nw)
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(27)
            p = s.plan()
            if nw == 0:
                assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_resident"]) == (2, 16, 11, 1), p     # 19 x 13 = 247 tiles of 176 rows
            else:
                assert p["tb_impl"] == 1
            assert_field(s.get_field(), want)
            d, MFL, MFR = s.flux()
            assert d == dor and np.array_equal(MFL, MFLo) and np.array_equal(MFR, MFRo)


def test_tall_tiles_whole_image_per_tile(pkg, oracle):
    """A stack of 128 x 128 images on tall tiles: ONE tile per image (16 waves x 8 rows, no halo beyond the walls), no
    neighbour to exchange with (rim stores and flags are skipped between passes), images stopping at different checks;
    iteration counts, Deff and fields against one-image-at-a-time oracle solves.  Also two tiles per image (128 x 200)."""
    for (nx, ny, B, R) in ((128, 128, 6, 8), (128, 200, 3, 8), (96, 64, 5, 4)):
        rng = np.random.default_rng(nx + ny + B)
        pixs = [rand_mask(rng, nx, ny, 0.35 + 0.08 * k) for k in range(B)]
        with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", R)
            s.set_image(np.stack(pixs))
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            res = s.solve(1e-3, 6000, check_every=200)
            got = s.get_field()
            p = s.plan()
            assert (p["tb_NW"], p["tb_R"], p["tb_resident"], p["tb_strips"]) == (16, R, 1, 1), p
            assert p["tb_chunks_per_image"] == (1 if ny <= 16 * R else 2)
        iters = set()
        for k in range(B):
            D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
            A, b = oracle.discretize(D, 0.0, 1.0)
            it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 6000,
                                                    check_every=200)
            assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
            assert_field(got[k * ny:(k + 1) * ny], x)
            iters.add(it)
        assert len(iters) > 1


def test_recommended_batch(pkg):
    """deff_recommended_batch: one slot per CU for images that are one tall tile, cell budgets otherwise."""
    per_cu = pkg.recommended_batch(128, 128, 100000)             # the CU count, rounded down to whole XCD rounds (256 on an MI355X)
    assert per_cu % 8 == 0 and 8 <= per_cu <= 1024
    assert pkg.recommended_batch(127, 200, 100000) == per_cu
    assert pkg.recommended_batch(128, 128, 10) == 10
    assert pkg.recommended_batch(128, 225, 100000) == (64 << 20) // (128 * 225)       # too tall for one tile: streaming stacks
    assert pkg.recommended_batch(1024, 1024, 1024) == 64
    assert pkg.recommended_batch(1024, 1024, 100) == 16
    assert pkg.recommended_batch(16384, 16384, 4) == 1
    with pkg.Solver(128, 128, nimg=pkg.recommended_batch(128, 128, 100000), kernel="matfree_tb") as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(64)
        p = s.plan()
        assert (p["tb_NW"], p["tb_R"], p["tb_resident"], p["tb_chunks_per_image"]) == (16, 8, 1, 1), p
        assert p["tb_blocks"] == per_cu                             # one image, one tile, one workgroup, one CU


def test_tall_tiles_symmetric_shortcut_is_verified_not_assumed(pkg, oracle):
    """Tall tiles do 7 lookups per row instead of 10 when the system is link-symmetric (a lane's second cell takes its W
    link from the first cell's E link, a row its N links from the S links of the row above).  That is checked on the
    device for the system at hand (k_links_symmetric; plan key tb_sym: 1 symmetric, 2 not): the native assembly passes;
    a host matrix whose E links were scaled without their W counterparts must not, and still gives its own oracle's bits;
    tb_sym = 2 switches the short-cut off."""
    rng = np.random.default_rng(314)
    nx, ny = 300, 260
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want = oracle.sweeps(A, b, x0, 27)
    for off in (0, 2):
        with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 8); s.set_tuning("tb_sym", off)
            s.set_image(pix)
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(27)
            p = s.plan()
            assert p["tb_NW"] == 16 and p["tb_sym"] == (1 if off == 0 else 0), p
            assert_field(s.get_field(), want)
    # the same matrix through the seam (harvested dictionary): still symmetric
    with pkg.Solver(nx, ny) as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 8)
        s.set_system(A, b, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(27)
        assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_NW"] == 16 and s.plan()["tb_sym"] == 1
        assert_field(s.get_field(), want)
    # E links scaled, W links not: few distinct rows (dictionary), no symmetry
    A2 = A.copy().reshape(ny, nx, 5)
    A2[:, :, 2] *= 1.0 + 2.0 ** -10
    A2 = A2.reshape(A.shape)
    want2 = oracle.sweeps(A2, b, x0, 27)
    assert not np.array_equal(want2, want)
    with pkg.Solver(nx, ny) as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 8)
        s.set_system(A2, b, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(27)
        p = s.plan()
        assert s.kernel_in_use() == "matfree_tb" and p["tb_NW"] == 16 and p["tb_sym"] == 2, p
        assert_field(s.get_field(), want2)
    # N links scaled: the vertical half of the check
    A3 = A.copy().reshape(ny, nx, 5)
    A3[:, :, 4] *= 1.0 - 2.0 ** -11
    A3 = A3.reshape(A.shape)
    want3 = oracle.sweeps(A3, b, x0, 27)
    with pkg.Solver(nx, ny) as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 16); s.set_tuning("tb_R", 8)
        s.set_system(A3, b, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(27)
        assert s.plan()["tb_sym"] == 2
        assert_field(s.get_field(), want3)


def test_tall_tiles_whole_images_beyond_one_per_cu(pkg, oracle):
    """Images that are one tall tile each wait for nobody, so a stack may hold more of them than the chip has CUs (the
    workgroups queue for the CUs): 260 images of 64 x 48, solved to their own stopping rules -- same iteration counts, Deff
    and fields as the streaming kernel, three of them against the oracle."""
    nx, ny, B = 64, 48, 260
    rng = np.random.default_rng(2600)
    pixs = np.stack([rand_mask(rng, nx, ny, 0.3 + 0.4 * rng.random()) for _ in range(B)])
    out = {}
    for form in ("tall", "streaming"):
        with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
            if form == "streaming":
                s.set_tuning("tb_impl", 1)
            s.set_image(pixs)
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            res = s.solve(1e-3, 4000, check_every=100)
            p = s.plan()
            if form == "tall":
                assert (p["tb_impl"], p["tb_NW"], p["tb_resident"], p["tb_strips"], p["tb_chunks_per_image"]) == (2, 16, 1, 1, 1), p
                assert p["tb_blocks"] == 264
            else:
                assert p["tb_impl"] == 1
            out[form] = ([(r.iters, r.deff_raw, r.conv) for r in res], s.get_field())
    assert out["tall"][0] == out["streaming"][0]
    assert np.array_equal(out["tall"][1], out["streaming"][1])
    assert len({t[0] for t in out["tall"][0]}) > 3
    for k in (0, 131, 259):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 4000, check_every=100)
        assert out["tall"][0][k] == (it, deff, conv)
        assert_field(out["tall"][1][k * ny:(k + 1) * ny], x)


# ---- link-symmetric tiles: 12 waves x R rows, symmetric matrix rows in registers (k_sweep_wgsym) ----

SYM_SHAPES = [(300, 200), (1030, 137), (600, 500), (250, 333), (2, 64), (97, 241), (122, 9), (1001, 333)]


@pytest.mark.parametrize("R", [4, 5])
@pytest.mark.parametrize("shape", SYM_SHAPES)
def test_sym_tiles_vs_oracle(pkg, oracle, shape, R):
    """k_sweep_wgsym<8, R>: ragged strips and row tiles, odd widths (padded column, b looked at everywhere), images shorter
    than one tile, a two-cell-wide image; 5 passes + 3 single sweeps, then 2 more passes in a second launch."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 13 + ny * 7 + R)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want1 = oracle.sweeps(A, b, x0, 43)
    want2 = oracle.sweeps(A, b, want1, 16)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12); s.set_tuning("tb_R", R)
        s.set_tuning("tb_wall_halo", (nx + ny + R) % 3)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(43)
        p = s.plan()
        tiles = p["tb_strips"] * p["tb_chunks_per_image"]
        if tiles <= 256 and (p["tb_LY"] >= 8 or p["tb_chunks_per_image"] == 1):
            assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"], p["tb_sym"]) == (2, 12, R, 8, 1, 1), p
            assert 1 <= p["tb_LY"] <= 12 * R - 16 and p["tb_LY"] * p["tb_chunks_per_image"] >= ny
            launches, _ = s.last_launches()
            assert launches == 1 + 3
        assert_field(s.get_field(), want1)
        s.sweeps(16)
        assert_field(s.get_field(), want2)


@pytest.mark.parametrize("T", [6, 4])
@pytest.mark.parametrize("R", [4, 5])
@pytest.mark.parametrize("shape", [(300, 200), (1030, 137), (250, 333), (2, 64), (97, 241), (1001, 333), (1152, 300), (1208, 150)])
def test_sym_tiles_with_passes_of_six_and_four_sweeps_vs_oracle(pkg, oracle, shape, R, T):
    """k_sweep_wgsym<6, R> and <4, R> (round 4): the same tiles with passes of SIX (FOUR) sweeps -- halo of 6 (4), 116 (120) owned
    columns, 12R - 12 (12R - 8) owned rows -- which is what lets images of 1101 ... 1172 (1208) columns stay on coefficient-resident
    tiles.  Ragged strips and row tiles, odd widths, a two-cell-wide image; 45 = 7 passes + 3 single sweeps (11 + 1), then
    13 = 2 passes + 1 (3 + 1); omega 2/3 and 1, both arithmetics."""
    nx, ny = shape
    rng = np.random.default_rng(nx * 11 + ny * 5 + R)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    for omega, kern, flavour, fma in ((2.0 / 3.0, 0, None, 0), (1.0, 1, None, 0), (2.0 / 3.0, 0, "fma", 1)):
        want1 = oracle.sweeps(A, b, x0, 45, kernel=kern, omega=omega, flavour=flavour)
        want2 = oracle.sweeps(A, b, want1, 13, kernel=kern, omega=omega, flavour=flavour)
        with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12); s.set_tuning("tb_R", R); s.set_tuning("tb_T", T); s.set_tuning("fma", fma)
            s.set_image(pix)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(45, omega)
            p = s.plan()
            assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"], p["tb_sym"]) == (2, 12, R, T, 1, 1), p
            assert 1 <= p["tb_LY"] <= 12 * R - 2 * T and p["tb_LY"] * p["tb_chunks_per_image"] >= ny
            assert p["tb_strips"] == (1 if nx <= 128 else 2 + (nx - 128 - 1) // (128 - 2 * T))
            assert s.last_launches() == (1 + 45 % T, T)
            assert np.array_equal(s.get_field(), want1)
            s.sweeps(13, omega)
            assert np.array_equal(s.get_field(), want2)


@pytest.mark.parametrize("shape_no,rows", [(1, (4, 4, 3)), (3, (5, 4, 4)), (4, (5, 5, 4))])
@pytest.mark.parametrize("size", [(300, 200), (1030, 137), (250, 333), (2, 64), (97, 241)])
def test_sym_tiles_with_rows_dealt_by_age(pkg, oracle, size, shape_no, rows):
    """k_sweep_wgsage<8, a, b, c> (round 4): the link-symmetric tiles with a row less for the younger waves of a SIMD -- tiles of
    4 (a + b + c) rows, each age on the pass loop instantiated for its own row count.  Ragged strips and row tiles, both
    arithmetics, omega 2/3 and 1; 45 = 5 passes + 5 single sweeps, then 13 more in a second launch."""
    nx, ny = size
    rng = np.random.default_rng(nx * 13 + ny * 7 + shape_no)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    owned = 4 * sum(rows) - 16
    for omega, kern, flavour, fma in ((2.0 / 3.0, 0, None, 0), (1.0, 1, None, 0), (2.0 / 3.0, 0, "fma", 1)):
        want1 = oracle.sweeps(A, b, x0, 45, kernel=kern, omega=omega, flavour=flavour)
        want2 = oracle.sweeps(A, b, want1, 13, kernel=kern, omega=omega, flavour=flavour)
        with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12); s.set_tuning("tb_sym_shape", shape_no); s.set_tuning("fma", fma)
            s.set_image(pix)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(45, omega)
            p = s.plan()
            assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"], p["tb_sym"], p["tb_aged"]) == (2, 12, rows[0], 8, 1, 1, 1), p
            assert 1 <= p["tb_LY"] <= owned and p["tb_LY"] * p["tb_chunks_per_image"] >= ny
            assert np.array_equal(s.get_field(), want1)
            s.sweeps(13, omega)
            assert np.array_equal(s.get_field(), want2)


def test_planner_takes_the_smallest_tile_shape_that_fits(pkg):
    """12-wave tiles, T = 8: 4/4/3, 4/4/4, 5/4/4, 5/5/4, 5/5/5 rows by age -- the first whose tiles are all on the chip."""
    for n, rows_first, tiles in ((896, 4, 256), (992, 5, 252), (1024, 5, 234), (1088, 5, 250)):
        with pkg.Solver(n, n) as s:
            s.synth_image(3, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(16)
            p = s.plan()
            assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"]) == (2, 12, rows_first, 8, 1), (n, p)
            assert p["tb_strips"] * p["tb_chunks_per_image"] == tiles, (n, p)
            assert p["tb_aged"] == (0 if n == 1088 else 1), (n, p)          # 1088^2 only fits with equal rows of 5


def test_planner_takes_passes_of_six_where_eight_do_not_fit(pkg, oracle):
    """1152^2 -- 9 x 128 columns, the worst point of round 3's size curve (648-672 G on tall tiles) -- is 27 x 11 = 297 tiles of the
    12-wave form at T = 8 and 24 x 10 = 240 at T = 6: the planner takes the latter on its own, a solve's 10 000-sweep intervals
    are 1 666 passes + 4 single sweeps, and the results are the oracle's; a size that fits at T = 8 keeps T = 8."""
    n = 1152
    pix = oracle.synth_mask(n, n, 5, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, want, _, _ = oracle.jacobi(A, b, oracle.linear_guess(n, n, 0.0, 1.0), D, 0.0, 1.0, 1e-9, 221, check_every=100)
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-9, 221, check_every=100)
        p = s.plan()
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"]) == (2, 12, 5, 6, 1), p
        assert p["tb_strips"] * p["tb_chunks_per_image"] == 240
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert np.array_equal(s.get_field(), want)
    with pkg.Solver(1024, 1024) as s:
        s.synth_image(1, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        assert (s.plan()["tb_T"], s.plan()["tb_NW"], s.plan()["tb_R"]) == (8, 12, 5)
    # 1200 columns: 11 strips at T = 6 need 128 + 10 x 116 = 1288 >= 1200 -> 11 strips x 25 row tiles = 275: no; T = 4: 10 strips
    # (128 + 9 x 120 = 1208) x ceil(1200 / 52) = 24 -> 240 tiles of 4 + 5 x 12 - 8 rows
    n = 1200
    pix = oracle.synth_mask(n, n, 6, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, want, _, _ = oracle.jacobi(A, b, oracle.linear_guess(n, n, 0.0, 1.0), D, 0.0, 1.0, 1e-9, 203, check_every=100)
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-9, 203, check_every=100)
        p = s.plan()
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"]) == (2, 12, 5, 4, 1), p
        assert p["tb_strips"] * p["tb_chunks_per_image"] == 240
        assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
        assert np.array_equal(s.get_field(), want)
    with pkg.Solver(1280, 1280) as s:                                  # too large for all three: tall tiles, T = 8
        s.synth_image(1, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        assert (s.plan()["tb_T"], s.plan()["tb_NW"]) == (8, 16)


def test_shorter_passes_on_a_stack_and_on_a_dictionary_system(pkg, oracle):
    """The planner's fall-back to passes of six / four sweeps beyond one 2-phase image: a STACK of two 1160 x 560 images (286 tiles
    at T = 8, 240 at T = 6) whose images stop at different checks, and a three-class system (rows from the harvested
    dictionary, checked link-symmetric) of 1190^2 (308 / 275 tiles at T = 8 / 6, 230 at T = 4); with impermeable solid the same
    image is not bit-symmetric and stays on tall tiles."""
    nx, ny, B = 1160, 560, 2
    rng = np.random.default_rng(77)
    pixs = [rand_mask(rng, nx, ny, 0.4 + 0.2 * k) for k in range(B)]
    with pkg.Solver(nx, ny, nimg=B) as s:
        s.set_image(np.stack(pixs))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-2, 900, check_every=100)
        got = s.get_field()
        p = s.plan()
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"]) == (2, 12, 5, 6, 1), p
        assert p["tb_strips"] * p["tb_chunks_per_image"] * B == 240
    iters = set()
    for k in range(B):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-2, 900, check_every=100)
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
        assert np.array_equal(got[k * ny:(k + 1) * ny], x)
        iters.add(it)
    assert len(iters) > 1
    n = 1190
    pix = np.where(rng.random((n, n)) < 0.3, 255, np.where(rng.random((n, n)) < 0.5, 120, 0)).astype(np.uint8)
    pix[0, :] = 255                                                    # solid top and bottom rows: no pore cell reaches across the wrap
    pix[-1, :] = 255
    # three classes with a permeable solid: DiscretizeMatrix2D on a three-valued D plane, rows harvested into the dictionary
    D3 = oracle.fill_D_3phase(pix, 1.0, 0.5, 30.0)
    A3, b3 = oracle.discretize(D3, 0.0, 1.0)
    want3 = oracle.sweeps(A3, b3, oracle.linear_guess(n, n, 0.0, 1.0), 23)
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_3phase(0.5, 1.0, 30.0, 0.0, 1.0, None)
        s.init_linear(0.0, 1.0)
        s.sweeps(23)
        p = s.plan()
        assert s.kernel_in_use() == "matfree_tb"
        assert (p["tb_impl"], p["tb_NW"], p["tb_R"], p["tb_T"], p["tb_resident"], p["tb_sym"]) == (2, 12, 5, 4, 1, 1), p
        assert np.array_equal(s.get_field(), want3)
    # impermeable solid: identity rows carry +0 links where their neighbours carry -0 -- not bit-symmetric, so the planner must
    # keep such a system off the link-symmetric tiles (tall tiles, looked-up rows)
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D3 = oracle.fill_D_3phase(pix, 1.0, 0.0, 30.0)
    with np.errstate(all="ignore"):
        A3, b3 = oracle.discretize(D3, 0.0, 1.0, grid=grid)
        want3 = oracle.sweeps(A3, b3, oracle.linear_guess(n, n, 0.0, 1.0), 23)
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 30.0, 0.0, 1.0, grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(23)
        p = s.plan()
        assert (p["tb_impl"], p["tb_NW"], p["tb_T"], p["tb_resident"], p["tb_sym"]) == (2, 16, 8, 1, 2), p
        assert_field(s.get_field(), want3)


@pytest.mark.parametrize("R", [4, 5])
def test_sym_tiles_omega_one_boundary_values_and_fma(pkg, oracle, R):
    nx, ny = 300, 250
    rng = np.random.default_rng(R + 40)
    pix = rand_mask(rng, nx, ny, 0.45)
    D = oracle.fill_D_2phase(pix, 2.0, 0.3)
    A, b = oracle.discretize(D, 2.0, -1.0)
    x0 = rng.random((ny, nx))
    for omega, kern in ((1.0, 1), (2.0 / 3.0, 0)):
        for flavour, fma in ((None, 0), ("fma", 1)):
            want = oracle.sweeps(A, b, x0, 25, kernel=kern, omega=omega, flavour=flavour)
            with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
                s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12); s.set_tuning("tb_R", R); s.set_tuning("fma", fma)
                s.set_image(pix)
                s.assemble_2phase(0.3, 2.0, 2.0, -1.0)
                s.set_field(x0)
                s.sweeps(25, omega)
                assert s.plan()["tb_NW"] == 12 and s.plan()["tb_resident"] == 1
                assert np.array_equal(s.get_field(), want)


def test_sym_tiles_are_taken_only_for_verified_symmetric_unguarded_systems(pkg, oracle, img00000):
    """The 12-wave form replaces a row's N links by the S links of the row above and a lane's second W link by its first E
    link, so it runs only where k_links_symmetric has verified that for the system at hand and no link needs the reference's
    non-zero test: the native assembly and the same matrix through the seam take it; a matrix with scaled E (or N) links,
    a zero-diffusivity phase (guard) and tb_sym = 2 do not -- and every case gives its own oracle's bits."""
    rng = np.random.default_rng(2718)
    nx, ny = 700, 600                                     # 7 strips x 19 row tiles: the 12-wave form is the planner's own choice
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    want = oracle.sweeps(A, b, x0, 27)
    for off in (0, 2):
        with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
            s.set_tuning("tb_sym", off)
            s.set_image(pix)
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(27)
            p = s.plan()
            assert (p["tb_NW"] == 12) == (off == 0) and p["tb_impl"] == 2, p       # the planner's own choice
            assert_field(s.get_field(), want)
    with pkg.Solver(nx, ny) as s:                                                 # harvested dictionary of the same matrix
        s.set_system(A, b, D, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(27)
        assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_NW"] == 12 and s.plan()["tb_sym"] == 1
        assert_field(s.get_field(), want)
    for plane, scale in ((2, 1.0 + 2.0 ** -10), (4, 1.0 - 2.0 ** -11)):
        A2 = A.copy().reshape(ny, nx, 5)
        A2[:, :, plane] *= scale
        A2 = A2.reshape(A.shape)
        want2 = oracle.sweeps(A2, b, x0, 27)
        assert not np.array_equal(want2, want)
        with pkg.Solver(nx, ny) as s:
            s.set_system(A2, b, D, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(27)
            p = s.plan()
            assert s.kernel_in_use() == "matfree_tb" and p["tb_NW"] != 12 and p["tb_sym"] == 2, p
            assert_field(s.get_field(), want2)
        with pkg.Solver(nx, ny) as s:                                             # ... even when the caller asks for it
            s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12); s.set_tuning("tb_R", 4)
            s.set_system(A2, b, D, 0.0, 1.0)
            s.set_field(x0)
            s.sweeps(27)
            assert s.plan()["tb_NW"] != 12
            assert_field(s.get_field(), want2)
    pix0 = np.tile(img00000, (2, 2))                                              # Ds = 0: -0.0 links, NaN cells, guarded kernels
    ny0, nx0 = pix0.shape
    D0 = oracle.fill_D_2phase(pix0, 1.0, 0.0)
    with np.errstate(all="ignore"):
        A0, b0 = oracle.discretize(D0, 0.0, 1.0)
        want0 = oracle.sweeps(A0, b0, oracle.linear_guess(nx0, ny0, 0.0, 1.0), 19)
    with pkg.Solver(nx0, ny0, kernel="matfree_tb") as s:
        s.set_image(pix0)
        s.assemble_2phase(0.0, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(19)
        assert s.plan()["tb_NW"] != 12
        assert_field(s.get_field(), want0)


def test_sym_tiles_stack_with_frozen_images_and_stream(pkg, oracle):
    """A stack on 12-wave tiles: images stop at different checks (their tiles leave the launch at once), and a stream
    refills the slots (new codes: the symmetry is re-verified); per image the oracle's sweeps, Deff, conv and field."""
    nx, ny, B = 250, 90, 4
    rng = np.random.default_rng(123)
    pixs = [rand_mask(rng, nx, ny, 0.35 + 0.1 * k) for k in range(7)]
    want = []
    for k in range(7):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        want.append(oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 6000, check_every=200))
    with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12)
        s.set_image(np.stack(pixs[:B]))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-3, 6000, check_every=200)
        got = s.get_field()
        p = s.plan()
        assert p["tb_NW"] == 12 and p["tb_resident"] == 1, p
    for k in range(B):
        it, deff, conv, x, _, _ = want[k]
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
        assert_field(got[k * ny:(k + 1) * ny], x)
    assert len({w[0] for w in want[:B]}) > 1
    with pkg.Solver(nx, ny, nimg=3, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2); s.set_tuning("tb_NW", 12)
        out = s.solve_stream(pixs, 1e-2, 1.0, 0.0, 1.0, 1e-3, 6000, check_every=200, want_fields=True)
        assert s.plan()["tb_NW"] == 12
    for k in range(7):
        it, deff, conv, x, _, _ = want[k]
        assert (out[k].iters, out[k].deff_raw, out[k].conv) == (it, deff, conv)
        assert_field(out[k].field, x)


def test_resident_fallback_in_stacks_and_streams(pkg, oracle):
    """The redo of an aborted resident interval with frozen images and with refilled slots around it: a stack whose images stop
    at different checks and a stream through 3 slots, both with a tile that never publishes from the first resident launch on
    (tb_debug_stall) -- the first interval is redone with one launch per pass, the rest of the solve runs that way, and every
    image has the oracle's sweeps, Deff, conv and field."""
    nx, ny, B = 250, 90, 4
    rng = np.random.default_rng(321)
    pixs = [rand_mask(rng, nx, ny, 0.35 + 0.1 * k) for k in range(7)]
    want = []
    for k in range(7):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        want.append(oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 6000, check_every=200))
    with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2)
        s.set_tuning("tb_debug_stall", 3)
        s.set_image(np.stack(pixs[:B]))
        s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        res = s.solve(1e-3, 6000, check_every=200)
        got = s.get_field()
        assert s.plan_value("tb_fallbacks") == 1 and s.plan()["tb_resident"] == 0
    for k in range(B):
        it, deff, conv, x, _, _ = want[k]
        assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
        assert_field(got[k * ny:(k + 1) * ny], x)
    with pkg.Solver(nx, ny, nimg=3, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2)
        s.set_tuning("tb_debug_stall", 2)
        out = s.solve_stream(pixs, 1e-2, 1.0, 0.0, 1.0, 1e-3, 6000, check_every=200, want_fields=True)
        assert s.plan_value("tb_fallbacks") == 1
    for k in range(7):
        it, deff, conv, x, _, _ = want[k]
        assert (out[k].iters, out[k].deff_raw, out[k].conv) == (it, deff, conv)
        assert_field(out[k].field, x)


def test_stream_that_ends_between_checks_settles_its_last_interval(pkg, oracle):
    """a stream whose last images run into MAX_ITER between two checks used to return with its last
    resident interval unchecked.  If that interval had given up, the images were handed out as they were, and the next
    deff_sweeps() on the context restored the stale restart copy over the caller's new field.  Here the last interval (49
    sweeps after the check at sweep 201) stalls: every image must still come out with the oracle's 250 sweeps, and a new
    field set on the same context afterwards must be swept from THAT field."""
    nx, ny = 250, 90
    rng = np.random.default_rng(99)
    pixs = [rand_mask(rng, nx, ny, 0.4 + 0.05 * k) for k in range(3)]
    sys_ = []
    for k in range(3):
        D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
        A, b = oracle.discretize(D, 0.0, 1.0)
        sys_.append((A, b, D))
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    want = [oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-12, 250, check_every=200) for A, b, D in sys_]
    with pkg.Solver(nx, ny, nimg=3, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 2)
        # the stream's resident launches: 199 sweeps between the checks at sweeps 1 and 201 (one launch), then the 49 sweeps up
        # to MAX_ITER (the second launch): that one stalls
        s.set_tuning("tb_debug_stall", 2)
        s.set_tuning("tb_debug_stall_skip", 1)
        out = s.solve_stream(pixs, 1e-2, 1.0, 0.0, 1.0, 1e-12, 250, check_every=200, want_fields=True)
        assert s.plan_value("tb_fallbacks") == 1
        for k in range(3):
            it, deff, conv, x, _, _ = want[k]
            assert it == 250 and (out[k].iters, out[k].deff_raw, out[k].conv) == (it, deff, conv)
            assert_field(out[k].field, x)
        # the same context, a new field: nothing of the stream's last interval may come back
        xs = 0.25 + 0.5 * x0
        s.set_field(np.tile(xs, (3, 1)))
        s.sweeps(19)
        got = s.get_field()
        for k in range(3):
            A, b, _ = sys_[k]
            assert_field(got[k * ny:(k + 1) * ny], oracle.sweeps(A, b, xs, 19))


def test_cooperative_launch_mode_is_gone(pkg):
    """tb_launch = 2 (hipLaunchCooperativeKernel) crashed in the ROCm runtime's teardown with several launching threads and is
    no longer part of the tuning surface: the key takes 0 or 1."""
    with pkg.Solver(64, 64) as s:
        s.set_tuning("tb_launch", 1)
        s.set_tuning("tb_launch", 0)
        with pytest.raises(pkg.DeffError, match="tb_launch takes"):
            s.set_tuning("tb_launch", 2)
