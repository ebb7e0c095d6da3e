"""Pins the CPU oracle (oracle/deff_oracle.c) before anything trusts it.

The reference has no test suite and its host loop cannot be built in this image
(its host-only functions and its kernels can: tests/test_ref_host.py,
tests/test_ref_kernel.py), so the pins here are (a) the reference outputs the survey stage recorded (SURVEY.md 6/8c,
BASELINE.md 2; copied in tests/golden/reference_recorded.json), (b) the
analytic known-answer cases of the reference's documentation (doc 5.3, with
the sweep counts the reference's own stopping rule produced), and (c) the
structural invariants of the discretisation.
All CPU; no GPU needed.
"""
import numpy as np
import pytest


def _stripe_mask(n, eps, series):
    k = int(round(eps * n))
    pix = np.full((n, n), 255, dtype=np.uint8)
    if series:
        pix[:, :k] = 0        # fluid columns first: phases in series along x
    else:
        pix[:k, :] = 0        # fluid rows first: phases in parallel
    return pix


def test_synthetic_first_check_deff_bit_exact(oracle, recorded):
    """Generator + D fill + assembly + one sweep + flux, against the reference's value."""
    rec = recorded["synthetic_first_check_deff"]
    for n in (128, 1024):
        pix = oracle.synth_mask(n, n, 12345, 0)
        D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
        A, b = oracle.discretize(D, 0.0, 1.0)
        x1 = oracle.sweeps(A, b, oracle.linear_guess(n, n, 0.0, 1.0), 1)
        deff, _, _ = oracle.flux_deff(x1, D, 0.0, 1.0)
        assert deff == rec[str(n)], (n, repr(deff), rec[str(n)])


def test_img00000_end_to_end_both_builds(oracle, recorded, img00000):
    """Config #1: 110 001 sweeps and both recorded Deff values, each reproduced to
    the last digit by one oracle build flavour (contraction off / allowed)."""
    rec = recorded["img00000_2phase_batch"]
    assert oracle.porosity(img00000) == rec["porosity"]
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = oracle.linear_guess(128, 128, 0.0, 1.0)
    got = {}
    fields = {}
    for flavour in (None, "fma"):
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, x0, D, 0.0, 1.0, 1e-6, 500000, flavour=flavour)
        assert it == rec["iters"]
        got[flavour] = (deff, conv)
        fields[flavour] = x
    deffs = {got[None][0], got["fma"][0]}
    assert deffs == {rec["deff_build_a"], rec["deff_build_b"]}, deffs
    assert rec["conv_build_a"] in (got[None][1], got["fma"][1])
    rel = np.linalg.norm(fields[None] - fields["fma"]) / np.linalg.norm(fields[None])
    assert abs(rel - rec["field_rel_l2_between_builds"]) < 1e-16, rel
    # committed golden field = contraction-off flavour
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "img00000_field.npy"))
    assert np.array_equal(gold, fields[None])


def test_two_phase_Ds0_gives_nan_after_one_sweep(oracle, recorded, img00000):
    D = oracle.fill_D_2phase(img00000, 1.0, 0.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0)
        it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(128, 128, 0.0, 1.0), D, 0.0, 1.0,
                                                1e-6, 500000)
    assert it == recorded["two_phase_Ds0"]["iters"]
    assert np.isnan(deff)


@pytest.mark.parametrize("idx", range(9))
def test_analytic_parallel(oracle, recorded, idx):
    """doc 5.3 eq. (7): Deff = eps*Df + (1-eps)*Ds; exact in the minimum 10 001 sweeps."""
    case = recorded["analytic_100x100_tol1e-6"]["parallel"][idx]
    eps, Ds = case["eps"], case["Ds"]
    pix = _stripe_mask(100, eps, series=False)
    D = oracle.fill_D_2phase(pix, 1.0, Ds)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(100, 100, 0.0, 1.0), D, 0.0, 1.0,
                                            1e-6, 5000000)
    assert it == case["sweeps"]
    exact = eps * 1.0 + (1 - eps) * Ds
    assert abs(deff - exact) / exact <= max(1e-12, 2 * case["rel_err_max"])


@pytest.mark.parametrize("idx", [0, 1, 3, 4])
def test_analytic_series(oracle, recorded, idx):
    """doc 5.3 eq. (8): Deff = (eps/Df + (1-eps)/Ds)^-1; the sweep count at which the
    reference's stopping rule fires is itself a golden value."""
    case = recorded["analytic_100x100_tol1e-6"]["series"][idx]
    eps, Ds = case["eps"], case["Ds"]
    pix = _stripe_mask(100, eps, series=True)
    D = oracle.fill_D_2phase(pix, 1.0, Ds)
    A, b = oracle.discretize(D, 0.0, 1.0)
    it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(100, 100, 0.0, 1.0), D, 0.0, 1.0,
                                            1e-6, 5000000)
    assert it == case["sweeps"]
    exact = 1.0 / (eps / 1.0 + (1 - eps) / Ds)
    rel = abs(deff - exact) / exact
    assert rel < 5e-6
    assert 0.5 * case["rel_err"] < rel < 2.0 * case["rel_err"]


def test_assembly_invariants(oracle):
    """E/W and S/N links are exactly symmetric, the diagonal is
    minus the sum of the links plus the wall terms, b lives on the walls only."""
    rng = np.random.default_rng(7)
    for (nx, ny) in [(8, 8), (16, 12), (33, 17)]:
        pix = np.where(rng.random((ny, nx)) < 0.5, 0, 255).astype(np.uint8)
        D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
        A, b = oracle.discretize(D, 0.25, 0.75)
        A2 = A.reshape(ny, nx, 5)
        assert np.array_equal(A2[:, :-1, 2], A2[:, 1:, 1])      # E of p == W of p+1
        assert np.array_equal(A2[:-1, :, 3], A2[1:, :, 4])      # S of p == N of p+nx
        assert np.all(A2[:, 0, 1] == 0) and np.all(A2[:, -1, 2] == 0)
        assert np.all(A2[0, :, 4] == 0) and np.all(A2[-1, :, 3] == 0)
        b2 = b.reshape(ny, nx)
        assert np.all(b2[:, 1:-1] == 0) and np.all(b2[:, 0] > 0) and np.all(b2[:, -1] > 0)
        wall = np.zeros((ny, nx))
        wall[:, 0] = D[:, 0] * (1.0 / ny) / ((1.0 / nx) / 2)
        wall[:, -1] = D[:, -1] * (1.0 / ny) / ((1.0 / nx) / 2)
        resid = A2[:, :, 0] + A2[:, :, 1:].sum(axis=2) - wall
        assert np.abs(resid).max() <= 1e-12 * np.abs(A2[:, :, 0]).max()


def test_golden_small_cases_match_oracle(oracle, small_cases):
    """The committed vectors are what the current oracle produces (guards both)."""
    for name in ("s8x8", "s16x12", "s33x17"):
        pix = small_cases[name + "_pix"]
        Ds, Df, CL, CR = small_cases[name + "_par"]
        ny, nx = pix.shape
        D = oracle.fill_D_2phase(pix, Df, Ds)
        A, b = oracle.discretize(D, CL, CR)
        assert np.array_equal(A, small_cases[name + "_A"]) and np.array_equal(b, small_cases[name + "_b"])
        x0 = oracle.linear_guess(nx, ny, CL, CR)
        for k in (1, 2, 100):
            assert np.array_equal(oracle.sweeps(A, b, x0, k, kernel=0), small_cases[f"{name}_sor{k}"])
            assert np.array_equal(oracle.sweeps(A, b, x0, k, kernel=1), small_cases[f"{name}_v1_{k}"])
        # omega = 1 through the SOR form equals the V1 kernel bit for bit (finite x)
        assert np.array_equal(oracle.sweeps(A, b, x0, 100, kernel=0, omega=1.0), small_cases[f"{name}_v1_100"])
        Ai, bi = oracle.discretize(oracle.fill_D_2phase(pix, Df, 0.0), CL, CR, grid=small_cases[name + "_grid"])
        assert np.array_equal(Ai, small_cases[name + "_Aimp"]) and np.array_equal(bi, small_cases[name + "_bimp"])
        g = small_cases[name + "_grid"].ravel()
        assert np.all(Ai[g > 0, 0] == 1) and np.all(Ai[g > 0, 1:] == 0)


def test_mesh_amplification_and_guess(oracle):
    pix = np.array([[0, 255], [255, 0]], dtype=np.uint8)
    D = oracle.fill_D_2phase(pix, 2.0, 0.5, ampX=3, ampY=2)
    assert D.shape == (4, 6)
    assert np.all(D[:2, :3] == 2.0) and np.all(D[:2, 3:] == 0.5) and np.all(D[2:, :3] == 0.5)
    x = oracle.linear_guess(6, 4, 0.25, 0.75)
    assert x[0, 0] == 0.25 and np.allclose(x[:, 3], 0.5) and np.all(np.diff(x, axis=0) == 0)


def test_img00000_3phase_as_shipped(oracle, recorded, img00000):
    """The configuration the reference's shipped input.txt selects (3 phases, Ds = 0, DCG
    continuation): FloodFill with its seeding quirk, DiscretizeMatrix2D_ImpSolid,
    JacobiGPUPreCond stages and the final JacobiGPU -- every recorded number reproduced."""
    rec = recorded["img00000_3phase_as_shipped"]
    o = rec["options"]
    with np.errstate(all="ignore"):
        r = oracle.solve_3phase(img00000, o["Ds"], o["Df"], o["Dg"], o["CL"], o["CR"], o["tol"], o["max_iter"])
    assert r["stage_sweeps"] == rec["stage_sweeps"] and sum(r["stage_sweeps"]) == rec["total_sweeps"]
    assert r["deff"] == rec["deff"]
    assert r["conv"] == rec["conv"]
    assert r["SVF"] == rec["SVF"] and r["LVF"] == rec["LVF"]


def test_floodfill_semantics(oracle):
    """Periodic in rows, not in columns; unreachable non-solid cells become 2; the right
    column is seeded iff the top-left cell is solid (reference quirk, cuh:601)."""
    g = np.ones((5, 6), dtype=np.uint32)
    g[2, :] = 0                       # an open channel in row 2 ...
    g[0, 3] = 0                       # ... an isolated pocket ...
    g[4, 0] = 0                       # ... and a left-wall cell that wraps to row 0 col 0 (solid)
    out, path = oracle.floodfill(g)
    assert path                      # top-left solid: right column seeded, PathFlag raised
    assert out[0, 3] == 2 and out[2, 2] == 0 and out[4, 0] == 0
    g2 = g.copy()
    g2[0, 0] = 0                      # top-left fluid: no right-column seeding
    g2[2, 3] = 1                      # break the channel
    out2, path2 = oracle.floodfill(g2)
    assert not path2
    assert out2[2, 4] == 2 and out2[2, 5] == 2 and out2[2, 1] == 0
    g3 = np.ones((4, 4), dtype=np.uint32)
    g3[0, 0] = 0
    g3[3, 0] = 0                      # reached through the periodic wrap from (0,0)
    g3[3, 1] = 0
    out3, _ = oracle.floodfill(g3)
    assert out3[3, 1] == 0
