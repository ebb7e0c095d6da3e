"""Residual() (Deff2D.cuh:451-494): the oracle's restatement checked against independent evaluations (CPU), and the HIP
reduction (deff_residual / deff_residual_D, kernels_residual.hpp) against the oracle (GPU).

Nothing in /root/reference holds a value of Residual() -- the function is dead code there (call sites commented out at
cuh:1121 and cuh:1266): PARITY UNPINNED beyond the restatement.  What pins the restatement here: (i) the exact discrete
solution of a uniform medium has residual ~0; (ii) on a square mesh (dx = dy) a cell's imbalance qW - qE + qN - qS is, up to
rounding, (A x - b) of the reference's own assembly, so mean|A x - b| must agree to ~1e-12 -- an evaluation that shares no
code with oracle_residual; (iii) a vectorised numpy restatement of the same expressions.

Bar for the HIP path (oracle_binding.assert_residual): every cell's term follows the reference's arithmetic and only the order
of the sum differs (wave-level tree instead of serial row-major).  The serial double sum is the LESS accurate of the two -- it
drifts from the exactly added terms by up to n * 2^-53 (observed 2e-12 at 512^2 on a rough medium) -- so the comparison is
(a) within 1e-13 of the oracle's per-cell doubles added in long double, which is what shows that the CELLS are right, and
(b) within max(1e-12, n * 2^-53) of the oracle's serial sum; plus bit-identical from call to call."""
import numpy as np
import pytest


def numpy_residual(x, D, CL, CR):
    """The expressions of cuh:451-494 vectorised (independent of oracle/deff_oracle.c)."""
    ny, nx = x.shape
    dx, dy = 1.0 / nx, 1.0 / ny
    w = dx / 2

    def H(a, b):
        with np.errstate(divide="ignore"):
            return (w + w) / (w / a + w / b)
    qW = np.empty_like(x); qE = np.empty_like(x); qN = np.zeros_like(x); qS = np.zeros_like(x)
    qW[:, 0] = dy / (dx / 2) * D[:, 0] * (x[:, 0] - CL)
    qW[:, 1:] = dy / dx * H(D[:, 1:], D[:, :-1]) * (x[:, 1:] - x[:, :-1])
    qE[:, -1] = dy / (dx / 2) * D[:, -1] * (CR - x[:, -1])
    qE[:, :-1] = dy / dx * H(D[:, :-1], D[:, 1:]) * (x[:, 1:] - x[:, :-1])
    qS[:-1] = dy / dx * H(D[1:], D[:-1]) * (x[1:] - x[:-1])
    qN[1:] = dy / dx * H(D[:-1], D[1:]) * (x[1:] - x[:-1])
    return float(np.abs(qW - qE + qN - qS).sum() / (nx * ny))


def test_oracle_residual_of_the_exact_uniform_solution_is_zero(oracle):
    nx, ny = 40, 24
    x = (np.arange(nx) + 0.5) / nx * np.ones((ny, 1)) * 0.5 + 0.25      # exact for CL = 0.25, CR = 0.75, any uniform D
    assert oracle.residual(x, np.full((ny, nx), 3.0), 0.25, 0.75) < 1e-15


@pytest.mark.parametrize("nx,ny", [(16, 12), (33, 17), (64, 64), (2, 2), (5, 3)])
def test_oracle_residual_vs_numpy_restatement(oracle, nx, ny):
    rng = np.random.default_rng(nx * 100 + ny)
    pix = np.where(rng.random((ny, nx)) < 0.5, 0, 255).astype(np.uint8)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    x = rng.random((ny, nx))
    a, b = oracle.residual(x, D, 0.0, 1.0), numpy_residual(x, D, 0.0, 1.0)
    assert abs(a - b) <= 1e-13 * abs(a)


def test_oracle_residual_is_the_matrix_residual_on_a_square_mesh(oracle):
    """dx = dy: qW - qE + qN - qS = (A x - b)_p with the reference's own A, b (cuh:815-902), so mean|A x - b| -- computed
    from the assembly, not from Residual()'s expressions -- must agree, during a solve and for a 3-class image."""
    n = 48
    pix = oracle.synth_mask(n, n, 7, 0)
    for D in (oracle.fill_D_2phase(pix, 1.0, 1e-3),
              oracle.fill_D_3phase((np.arange(n * n).reshape(n, n) * 37 % 256).astype(np.uint8), 1.0, 0.5, 30.0)):
        A, b = oracle.discretize(D, 0.0, 1.0)
        x = oracle.linear_guess(n, n, 0.0, 1.0)
        for sweeps in (0, 10, 500):
            x = oracle.sweeps(A, b, x, sweeps) if sweeps else x
            xf = x.ravel()
            Ax = A[:, 0] * xf
            Ax[1:] += A[1:, 1] * xf[:-1]
            Ax[:-1] += A[:-1, 2] * xf[1:]
            Ax[:-n] += A[:-n, 3] * xf[n:]
            Ax[n:] += A[n:, 4] * xf[:-n]
            want = float(np.abs(Ax - b).sum() / (n * n))
            got = oracle.residual(x, D, 0.0, 1.0)
            assert abs(got - want) <= 1e-11 * want, (sweeps, got, want)


# ------------------------------------------------------------------ GPU: the HIP reduction against the oracle -----------

@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p




@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny", [(2, 2), (3, 5), (16, 12), (33, 17), (97, 41), (128, 8), (130, 9), (256, 256), (257, 33), (1030, 37), (514, 100)])
def test_residual_2phase_vs_oracle(pkg, oracle, nx, ny):
    """Ragged tiles (one lane ... several strips), odd widths (padded arrays), the first sweeps of a solve; walls 0.25 / 0.75;
    the class kernel (no D plane) and the plane kernel give the oracle's value, and the same bits on a second call."""
    rng = np.random.default_rng(nx * 1000 + ny)
    pix = np.where(rng.random((ny, nx)) < 0.45, 0, 255).astype(np.uint8)
    CL, CR = 0.25, 0.75
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, CL, CR)
    x = oracle.linear_guess(nx, ny, CL, CR)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, CL, CR)
        s.init_linear(CL, CR)
        for sweeps in (0, 1, 26):
            if sweeps:
                s.sweeps(sweeps)
                x = oracle.sweeps(A, b, x, sweeps)
            got = s.residual()
            oracle.assert_residual(got, x, D, CL, CR)
            assert all(s.residual() == got for _ in range(20))           # deterministic
            oracle.assert_residual(s.residual(D, CL, CR), x, D, CL, CR)
        # any field, not only iterates
        f = rng.random((ny, nx)) * 3 - 1
        s.set_field(f)
        oracle.assert_residual(s.residual(), f, D, CL, CR)


@pytest.mark.gpu
@pytest.mark.parametrize("ampX,ampY", [(2, 1), (1, 3), (3, 2)])
def test_residual_with_mesh_amplification(pkg, oracle, ampX, ampY):
    W, H = 43, 21
    rng = np.random.default_rng(5)
    pix = np.where(rng.random((H, W)) < 0.5, 0, 255).astype(np.uint8)
    D = oracle.fill_D_2phase(pix, 2.0, 1e-2, ampX, ampY)
    ny, nx = D.shape
    A, b = oracle.discretize(D, 0.0, 1.0)
    x = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 9)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix, ampX, ampY)
        s.assemble_2phase(1e-2, 2.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(9)
        oracle.assert_residual(s.residual(), x, D, 0.0, 1.0)


@pytest.mark.gpu
def test_residual_img00000_converged_field(pkg, oracle, img00000):
    """Config #1: the reference's own image, the field of the 110 001-sweep solve (committed golden): the residual the
    reference would have printed at its last check had the call at cuh:1266 not been commented out."""
    import os
    from conftest import GOLDEN
    x = np.load(os.path.join(GOLDEN, "img00000_field.npy"))
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    want = oracle.residual(x, D, 0.0, 1.0)
    with pkg.Solver(128, 128) as s:
        s.set_image(img00000)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x)
        got = s.residual()
    oracle.assert_residual(got, x, D, 0.0, 1.0)
    assert 0 < want < 1e-6


@pytest.mark.gpu
def test_residual_3phase_with_impermeable_solid(pkg, oracle):
    """3 pixel classes, Ds = 0 (every face of a solid cell has conductance exactly 0 through H's inf), ImpSolid rows from the
    flood-filled Grid: the residual reads pixels and the field only."""
    nx, ny = 150, 64
    rng = np.random.default_rng(11)
    pix = rng.choice(np.array([0, 30, 120, 199, 201, 255], dtype=np.uint8), size=(ny, nx), p=[0.25, 0.1, 0.25, 0.1, 0.1, 0.2])
    pix[0] = pix[-1] = 255        # FloodFill wraps top <-> bottom (cuh:641-664), the matrix does not: keep pockets from hiding there
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
    x = oracle.sweeps(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), 40)
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid=grid)
        s.init_linear(0.0, 1.0)
        s.sweeps(40)
        assert np.isfinite(x).all() and np.array_equal(s.get_field(), x)
        assert np.isfinite(oracle.residual(x, D, 0.0, 1.0))
        oracle.assert_residual(s.residual(), x, D, 0.0, 1.0)
        oracle.assert_residual(s.residual(D, 0.0, 1.0), x, D, 0.0, 1.0)


@pytest.mark.gpu
def test_residual_of_a_stack_is_per_image(pkg, oracle):
    nx, ny, B = 250, 90, 5
    pix = np.stack([oracle.synth_mask(nx, ny, 77, k) for k in range(B)])
    with pkg.Solver(nx, ny, nimg=B) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(17)
        got = s.residual()
        f = s.get_field()
    assert got.shape == (B,)
    for k in range(B):
        D = oracle.fill_D_2phase(pix[k], 1.0, 1e-3)
        oracle.assert_residual(got[k], f[k * ny:(k + 1) * ny], D, 0.0, 1.0)


@pytest.mark.gpu
def test_residual_from_the_progress_callback_and_after_assemble_from_D(pkg, oracle):
    nx, ny = 96, 64
    pix = oracle.synth_mask(nx, ny, 3, 0)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    seen = []
    with pkg.Solver(nx, ny) as s:
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.set_progress(lambda k, d, ch: seen.append((k, s.residual())))
        r = s.solve(1e-30, 401, check_every=200)
        assert r.iters == 401 and [k for k, _ in seen] == [0, 200, 400]
    x = oracle.linear_guess(nx, ny, 0.0, 1.0)
    done = 0
    for k, got in seen:
        x = oracle.sweeps(A, b, x, k + 1 - done)
        done = k + 1
        oracle.assert_residual(got, x, D, 0.0, 1.0)
    assert seen[0][1] > seen[1][1] > seen[2][1]
    # a system assembled from a D plane carries no pixel classes: deff_residual says so, deff_residual_D works
    with pkg.Solver(nx, ny) as s:
        s.assemble_from_D(D, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        with pytest.raises(pkg.DeffError, match="deff_residual_D"):
            s.residual()
        oracle.assert_residual(s.residual(D, 0.0, 1.0), oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0)
