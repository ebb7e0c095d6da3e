"""oracle/_ref/ref_loop -- the reference's host loop JacobiGPU (Deff2D.cuh:1163-1314) on the MI355X, from its own logic lines.

The loop interleaves plain C++ (declarations with the literals deffOld = 5, percentChange = 100, iterToCheck = 10000; the while
condition; the check condition; the wall-flux / Deff / relative-change evaluation; the counter; the outputs) with CUDA-runtime
calls that only move data.  oracle/ref_loop_probe.hip (ours) is a function of the same signature whose logic lines are the
reference's, #included verbatim as fragments cut from the file where it lies, and whose own lines are that data movement
through HIP.  With oracle/_ref/ref_host (assembly) and the kernels inside it, this is the reference's solve on this GPU except
for the drivers' image handling:

    reference's DiscretizeMatrix2D -> reference's JacobiGPU lines + updateX_SOR  ==  deff_assemble_2phase + deff_solve
    (iteration count, Deff at the last check, last signed change, final field, wall fluxes: bit for bit)

This closes the last restated piece of the path -- the stopping rule and the Deff evaluation (rows a11, a12) -- against the
reference's own statements.  What remains restated: the drivers (D fill from pixels, linear guess, continuation ramps), whose
lines sit inside functions that also open files and call the CUDA runtime."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ob.have_ref_loop() and ob.have_ref_host()), reason="oracle/_ref not built (needs /root/reference at build time)")]


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


def solve_hip(pkg, pix, Ds, Df, CL, CR, tol, max_iter, kernel="auto"):
    ny, nx = pix.shape
    with pkg.Solver(nx, ny, kernel=kernel) as s:
        s.set_image(pix)
        s.assemble_2phase(Ds, Df, CL, CR)
        s.init_linear(CL, CR)
        r = s.solve(tol, max_iter)
        return r, s.get_field()


def test_config1_through_the_references_own_loop(pkg, oracle, img00000, recorded, tmp_path):
    """Config #1 with the reference's code on every line of the solve: its assembly, its loop logic, its kernel -- 110 001
    iterations, Deff 0.18286248993335824 (the value at its last check), conv, field, wall fluxes: the oracle's, the recorded
    ones and deff_solve's, bit for bit."""
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    x0 = oracle.linear_guess(128, 128, 0.0, 1.0)
    it, deff, conv, x, MFL, MFR, ms = ob.ref_jacobi(A, b, x0, D, 0.0, 1.0, 1e-6, 500000, tmpdir=tmp_path)
    rec = recorded["img00000_2phase_batch"]
    assert it == 110001 == rec["iters"] and deff == rec["deff_build_b"]
    oit, odeff, oconv, ox, oL, oR = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-6, 500000)
    assert (it, deff, conv) == (oit, odeff, oconv) and np.array_equal(x, ox) and np.array_equal(MFL, oL) and np.array_equal(MFR, oR)
    r, got = solve_hip(pkg, img00000, 1e-3, 1.0, 0.0, 1.0, 1e-6, 500000)
    assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
    assert np.array_equal(got, x) and np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)
    print(f"reference loop: {it} iterations in {ms / 1e3:.2f} s of its own timing window")


@pytest.mark.parametrize("tol,max_iter", [(1e-6, 1), (1e-6, 9999), (1e-6, 10000), (1e-6, 10001), (1e-6, 10002), (1e-6, 15000),
                                          (1e-6, 20001), (1e-1, 500000), (1e-2, 500000), (99.0, 500000), (100.0, 500000), (1e3, 500000)])
def test_stopping_rule_edges_against_the_references_own_loop(pkg, oracle, tol, max_iter, tmp_path):
    """The rule's corners on a 96 x 64 image: MAX_ITER before / on / after a check (the Deff reported is the value at the LAST
    CHECK, not of the final field), loose tolerances that stop at the first or second check (the first compares with the
    literal 5), and tolerances >= the initial 100 that admit no sweep at all."""
    nx, ny = 96, 64
    pix = oracle.synth_mask(nx, ny, 3, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-2)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    it, deff, conv, x, MFL, MFR, _ = ob.ref_jacobi(A, b, x0, D, 0.0, 1.0, tol, max_iter, tmpdir=tmp_path)
    oit, odeff, oconv, ox, _, _ = oracle.jacobi(A, b, x0, D, 0.0, 1.0, tol, max_iter)
    assert (it, deff, conv) == (oit, odeff, oconv), (it, oit)
    r, got = solve_hip(pkg, pix, 1e-2, 1.0, 0.0, 1.0, tol, max_iter)
    assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv), (r.iters, it)
    if it == 0:
        # The one place where this build deliberately differs from the reference's bytes: a solve that admits no sweep (a
        # tolerance >= the initial "change" of 100, i.e. 10 000 %).  The reference still copies d_x_vec back (cuh:1300), the
        # buffer its kernel WOULD have written -- zeros from initializeGPU on an image's first call -- and so hands back a zero
        # field; oracle and library leave the caller's guess untouched.  Count, Deff (the initial 1) and conv agree.
        assert not x.any() and np.array_equal(ox, x0) and np.array_equal(got, x0)
        return
    assert np.array_equal(x, ox)
    assert np.array_equal(got, x)


@pytest.mark.parametrize("kernel", ["matfree_tb", "explicit"])
def test_1024_first_checks_against_the_references_own_loop(pkg, oracle, kernel, tmp_path):
    """Config #2's image through three checks (20 001 iterations): the reference's loop against deff_solve on the resident tiles
    and on the explicit kernel."""
    n = 1024
    pix = oracle.synth_mask(n, n, 12345, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
    it, deff, conv, x, MFL, MFR, _ = ob.ref_jacobi(A, b, oracle.linear_guess(n, n, 0.0, 1.0), D, 0.0, 1.0, 1e-9, 20001, tmpdir=tmp_path)
    r, got = solve_hip(pkg, pix, 1e-3, 1.0, 0.0, 1.0, 1e-9, 20001, kernel=kernel)
    assert it == 20001 and (r.iters, r.deff_raw, r.conv) == (it, deff, conv)
    assert np.array_equal(got, x) and np.array_equal(r.MFL, MFL) and np.array_equal(r.MFR, MFR)


def test_3phase_as_shipped_through_the_references_own_loops(pkg, oracle, img00000, recorded, tmp_path):
    """Row a14 and the configuration the reference ships (3 phases, Ds 0, Df 1, Dg 1 237 500, tol 1e-5) on 00000.jpg, with the
    reference's code on every numeric line: its flood fill, its 3-class D fill, its DiscretizeMatrix2D_ImpSolid, six stages of its
    JacobiGPUPreCond (DCG = 10 ... 1e6, tolerance x 10, MAX_ITER 1e6) and its JacobiGPU, each warm-started from the previous
    field (x_vec is in / out) -- only the ramp's control flow (cuh:1492-1549: g = 10; g < DCG; g *= 10) is written here.
    Stage counts [80 001, 10 001 x 5, 10 001], Deff 224673.61044289195 (/ Df), conv: the recorded numbers, the oracle's and the
    library's (batch.solve_image_3phase), bit for bit; the final fields too."""
    from effectivediffusivityfvm_amd import batch
    rec = recorded["img00000_3phase_as_shipped"]
    o = rec["options"]
    Ds, Df, Dg, CL, CR, tol, max_iter = o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"], o["max_iter"]
    grid = ob.ref_floodfill((img00000 > 200).astype(np.uint32), tmpdir=tmp_path)
    x = ob.ref_linear_guess(128, 128, CL, CR, tmpdir=tmp_path)
    stages = []
    g = 10.0
    with np.errstate(all="ignore"):
        while g < Dg:                                                                # cuh:1492
            D = ob.ref_fill_D(img00000, Ds, Df, g, phases=3, tmpdir=tmp_path)
            A, b = ob.ref_discretize(D, CL, CR, grid=grid, tmpdir=tmp_path)
            it, _, _, x, _, _, _ = ob.ref_jacobi(A, b, x, D, CL, CR, tol * 10, 1000000, tmpdir=tmp_path, precond=True)   # cuh:1503, 1539
            stages.append(it)
            g = g * 10                                                               # cuh:1548
        D = ob.ref_fill_D(img00000, Ds, Df, Dg, phases=3, tmpdir=tmp_path)
        A, b = ob.ref_discretize(D, CL, CR, grid=grid, tmpdir=tmp_path)
        it, deff, conv, x, _, _, _ = ob.ref_jacobi(A, b, x, D, CL, CR, tol, max_iter, DCfluid=Df, tmpdir=tmp_path)      # cuh:1590
    stages.append(it)
    assert stages == rec["stage_sweeps"]
    assert deff / Df == rec["deff"] and conv == rec["conv"]
    with np.errstate(all="ignore"):
        want = oracle.solve_3phase(img00000, Ds, Df, Dg, CL, CR, tol, max_iter)
    assert want["stage_sweeps"] == stages and want["deff"] == deff / Df and want["conv"] == conv
    assert np.array_equal(want["field"], x, equal_nan=True)
    with pkg.Solver(128, 128) as s:
        got = batch.solve_image_3phase(s, img00000, Ds, Df, Dg, CL, CR, tol, max_iter)
        field = s.get_field()
    assert got["stage_sweeps"] == stages and got["deff"] == deff / Df and got["conv"] == conv
    assert np.array_equal(field, x, equal_nan=True)


def test_single_sim_dcf_ramp_through_the_references_own_loop(oracle, img00000, tmp_path):
    """SingleSim's DCF continuation (cuh:1759-1817) for Df = 1e4 on 00000.jpg, MaxIter 30 001: stages DCF = 100 and 1e4, each the
    drivers' own D fill, the reference's assembly and its JacobiGPU, warm-started through x_vec; the ramp's control flow is the
    oracle binding's restatement (solve_single_2phase_ramp), whose per-stage counts, Deff / DCF, conv and final field must come out
    of the reference's code too."""
    want = oracle.solve_single_2phase_ramp(img00000, 1e-3, 1e4, 0.0, 1.0, 1e-6, 30001)
    x = ob.ref_linear_guess(128, 128, 0.0, 1.0, tmpdir=tmp_path)
    got = []
    for DCF in (100.0, 1e4):                                                         # std::pow(100, count), clipped to Df
        D = ob.ref_fill_D(img00000, 1e-3, DCF, tmpdir=tmp_path)
        A, b = ob.ref_discretize(D, 0.0, 1.0, tmpdir=tmp_path)
        it, deff, conv, x, _, _, _ = ob.ref_jacobi(A, b, x, D, 0.0, 1.0, 1e-6, 30001, DCfluid=DCF, tmpdir=tmp_path)
        got.append((DCF, it, deff / DCF, conv))
    assert got == want["stages"]
    assert np.array_equal(x, want["field"])
