"""oracle/_ref/ref_kernel -- the reference's OWN sweep kernels on the MI355X.

updateX_SOR / updateX_V1 (Deff2D.cuh:69-118) are CUDA kernel language, which is HIP's kernel language: hipcc compiles their text
for gfx950 as it lies (oracle/Makefile target `ref`; the probe around them, oracle/ref_kernel_probe.hip, is ours and launches them
with the reference's grid of n/160 + 1 blocks of 160 threads and its x <- xNew copy after every launch).  Together with
oracle/_ref/ref_host (the reference's own assembly) this puts the reference's code on both sides of rows a5/a6 and a9/a10:

    reference's DiscretizeMatrix2D -> A, b -> reference's updateX_SOR x k   ==   this library's assembly + k sweeps, bit for bit

on every kernel form (explicit, matrix-free, temporally blocked streaming / workgroup tiles / resident).  The written-order
build (-ffp-contract=off) is the library's default arithmetic; the build with hipcc's default contraction is compared with the
library's "fma" mode.  What stays a restatement: the host loop's stopping rule and Deff evaluation (cuh:1232-1290: CUDA calls
throughout, not separable)."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ob.have_ref_kernel() and ob.have_ref_host()), reason="oracle/_ref not built (needs /root/reference at build time)")]


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("nx,ny", [(8, 8), (16, 12), (33, 17), (128, 128), (130, 70), (512, 512)])
def test_oracle_sweeps_equal_the_references_kernels(oracle, nx, ny, tmp_path):
    """a9 / a10: the oracle's sweep restatement against the reference's kernels run on the GPU: 1, 2 and 27 sweeps of
    updateX_SOR and updateX_V1 on the reference's own assembly -- bit for bit."""
    rng = np.random.default_rng(nx * 31 + ny)
    pix = np.where(rng.random((ny, nx)) < 0.5, 0, 255).astype(np.uint8)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    for which in (0, 1):
        for k in (1, 2, 27):
            want = ob.ref_sweeps(A, b, x0, k, which=which, tmpdir=tmp_path)
            got = oracle.sweeps(A, b, x0, k, kernel=which, omega=(2.0 / 3.0 if which == 0 else 1.0))
            assert np.array_equal(got, want), (which, k, rel_l2(got, want))


@pytest.mark.parametrize("kernel,tune", [("explicit", {}), ("scalar", {}), ("matfree", {}), ("matfree_tb", {"tb_impl": 1}),
                                         ("matfree_tb", {"tb_impl": 2, "tb_launch": 1}), ("matfree_tb", {"tb_impl": 2})])
@pytest.mark.parametrize("n", [128, 1024])
def test_hip_sweeps_equal_the_references_kernels(pkg, oracle, img00000, kernel, tune, n, tmp_path):
    """The product against the reference's code, end to end through assembly and sweeps: image -> (reference) D fill restated ->
    reference's DiscretizeMatrix2D -> 27 launches of the reference's updateX_SOR on the GPU, against image -> deff_assemble_2phase
    -> 27 sweeps of each kernel form (3 blocked passes + 3 single sweeps for the temporally blocked ones): the same field, bit
    for bit; omega = 1 against updateX_V1.  128^2 is the reference's own image (config #1), 1024^2 the synthetic config #2."""
    pix = img00000 if n == 128 else oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    x0 = oracle.linear_guess(n, n, 0.0, 1.0)
    for which, omega in ((0, 2.0 / 3.0), (1, 1.0)):
        want = ob.ref_sweeps(A, b, x0, 27, which=which, tmpdir=tmp_path)
        with pkg.Solver(n, n, kernel=kernel) as s:
            for k, v in tune.items():
                s.set_tuning(k, v)
            s.set_image(pix)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(27, omega)
            got = s.get_field()
        assert rel_l2(got, want) <= 1e-6                              # north star
        assert np.array_equal(got, want), (kernel, tune, which, rel_l2(got, want))


@pytest.mark.parametrize("n,form", [(1152, (12, 6)), (1200, (12, 4)), (1536, (16, 8))])
def test_mid_size_default_plans_equal_the_references_kernel(pkg, oracle, n, form, tmp_path):
    """The resident forms the planner picks between 1100^2 and 2300^2 -- 12-wave link-symmetric tiles with passes of six and of
    four sweeps, tall 16-wave tiles -- against 27 launches of the reference's updateX_SOR after the reference's own assembly."""
    pix = oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    want = ob.ref_sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 27, tmpdir=tmp_path)
    with pkg.Solver(n, n) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(27)
        p = s.plan()
        assert (p["tb_NW"], p["tb_T"]) == form and p["tb_resident"] == 1, p
        assert np.array_equal(s.get_field(), want)


def test_hip_3phase_sweeps_equal_the_references_kernels(pkg, oracle, tmp_path):
    """Three pixel classes, impermeable solid (Ds = 0), the reference's own flood fill and ImpSolid assembly, the reference's
    kernel with its non-zero link test (cuh:77) against the library's harvested dictionary + guarded kernels."""
    nx, ny = 150, 64
    rng = np.random.default_rng(11)
    pix = rng.choice(np.array([0, 30, 120, 199, 201, 255], dtype=np.uint8), size=(ny, nx), p=[0.25, 0.1, 0.25, 0.1, 0.1, 0.2])
    pix[0] = pix[-1] = 255
    grid = ob.ref_floodfill((pix > 200).astype(np.uint32), tmpdir=tmp_path)
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    A, b = ob.ref_discretize(D, 0.0, 1.0, grid=grid, tmpdir=tmp_path)
    want = ob.ref_sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 40, tmpdir=tmp_path)
    for kernel in ("explicit", "matfree_tb"):
        with pkg.Solver(nx, ny, kernel=kernel) as s:
            s.set_image(pix)
            s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid=grid)
            s.init_linear(0.0, 1.0)
            s.sweeps(40)
            assert np.array_equal(s.get_field(), want, equal_nan=True), kernel


def test_contracted_build_of_the_references_kernel_against_the_fma_mode(pkg, oracle, img00000, tmp_path):
    """hipcc's default (-ffp-contract=fast) on the reference's kernel text is the counterpart of nvcc's default -fmad=true -- the
    arithmetic the shipped CUDA binary most likely runs.  The library's "fma" mode was written after gcc's contraction of the
    oracle; this records how it relates to hipcc's contraction of the reference's own text: within 1e-12 relative L2 after 100
    sweeps (eight orders inside the north star's 1e-6), and bit-identical if the two compilers fuse the same products."""
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    x0 = oracle.linear_guess(128, 128, 0.0, 1.0)
    want = ob.ref_sweeps(A, b, x0, 100, fma=True, tmpdir=tmp_path)
    plain = ob.ref_sweeps(A, b, x0, 100, tmpdir=tmp_path)
    with pkg.Solver(128, 128) as s:
        s.set_tuning("fma", 1)
        s.set_image(img00000)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)                      # (fma mode also contracts the linear guess: compare from a common start)
        s.set_field(x0)
        s.sweeps(100)
        got = s.get_field()
    print(f"fma mode vs hipcc-contracted reference kernel: rel L2 {rel_l2(got, want):.3e}, bit-identical: {np.array_equal(got, want)}; "
          f"contracted vs written order of the reference kernel: rel L2 {rel_l2(want, plain):.3e}")
    assert rel_l2(got, want) <= 1e-12 and rel_l2(want, plain) <= 1e-12
    assert np.array_equal(got, want)                 # observed: hipcc fuses the reference's text exactly as the "fma" mode is written
    assert not np.array_equal(want, plain)           # ... and the two arithmetics do differ (5e-16 after 100 sweeps)


def test_4096_against_the_references_own_code(pkg, oracle, tmp_path):
    """Where the money is, with the reference on the other side: the benchmark image at 4096^2 -- the reference's own
    DiscretizeMatrix2D (671 MB of A), 27 launches of the reference's own updateX_SOR on the MI355X -- against
    deff_assemble_2phase + 27 sweeps of the timed kernel (3 temporally blocked passes of 8 + 3 single sweeps): the same 16.7 M
    doubles, bit for bit."""
    n = 4096
    pix = oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    want = ob.ref_sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 27, tmpdir=tmp_path)
    del A, b, D
    with pkg.Solver(n, n, kernel="matfree_tb") as s:
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(27)
        assert s.plan()["tb_T"] == 8 and s.plan()["tb_impl"] == 1
        got = s.get_field()
    assert rel_l2(got, want) <= 1e-6 and np.array_equal(got, want)


def test_config1_field_after_110001_launches_of_the_references_kernel(pkg, oracle, img00000, recorded, tmp_path):
    """Config #1 in full: the reference's image, the reference's assembly, 110 001 launches of the reference's updateX_SOR on the
    MI355X (the count at which the reference's stopping rule fires, reproduced by deff_solve) -- the field equals the committed
    golden field (generated by the oracle in an earlier round) and the field deff_solve leaves, bit for bit; the Deff evaluated
    from it by the oracle's restatement of cuh:1252-1263 is the recorded value."""
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "img00000_field.npy"))
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    want = ob.ref_sweeps(A, b, oracle.linear_guess(128, 128, 0.0, 1.0), 110001, tmpdir=tmp_path)
    assert np.array_equal(want, gold)
    with pkg.Solver(128, 128) as s:
        s.set_image(img00000)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        r = s.solve(1e-6, 500000)
        got = s.get_field()
    assert r.iters == 110001 == recorded["img00000_2phase_batch"]["iters"]
    assert np.array_equal(got, want)
    # the Deff of the reference's loop is the value at its LAST CHECK (sweep 110 001 = this field)
    assert oracle.flux_deff(want, D, 0.0, 1.0)[0] == r.deff_raw == recorded["img00000_2phase_batch"]["deff_build_b"]
