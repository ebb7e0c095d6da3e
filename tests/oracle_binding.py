"""ctypes binding of oracle/libdeff_oracle.so (test infrastructure only).

The oracle is the CPU restatement of the reference hot path; see the header of
oracle/deff_oracle.c.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None
_LIBS = {}

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib(flavour=None):
    """flavour None = the parity oracle (-ffp-contract=off); "fma" = the
    contraction-allowed build used only for golden pinning."""
    global _LIB
    if flavour is None and _LIB is not None:
        return _LIB
    if flavour in _LIBS:
        return _LIBS[flavour]
    name = "libdeff_oracle.so" if flavour is None else f"libdeff_oracle_{flavour}.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.oracle_synth_mask.argtypes = [_u8p, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
    L.oracle_synth_mask.restype = None
    L.oracle_porosity.argtypes = [_u8p, C.c_int, C.c_int]
    L.oracle_porosity.restype = C.c_double
    L.oracle_fill_D_2phase.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_double, _dp]
    L.oracle_fill_D_2phase.restype = None
    L.oracle_fill_D_3phase.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_double, C.c_double, _dp]
    L.oracle_fill_D_3phase.restype = None
    L.oracle_linear_guess.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double]
    L.oracle_linear_guess.restype = None
    L.oracle_whm.argtypes = [C.c_double] * 4
    L.oracle_whm.restype = C.c_double
    L.oracle_discretize_2d.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_double, C.c_double]
    L.oracle_discretize_2d.restype = None
    L.oracle_discretize_2d_impsolid.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                                C.c_double, C.c_double, C.c_double, _u32p]
    L.oracle_discretize_2d_impsolid.restype = None
    L.oracle_sweeps.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_long, C.c_int, C.c_double]
    L.oracle_sweeps.restype = None
    L.oracle_flux_deff.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                   _dp, _dp]
    L.oracle_flux_deff.restype = C.c_double
    L.oracle_jacobi.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                C.c_double, C.c_long, C.c_long, _dp, _dp, _dp, C.c_int, C.c_double,
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.oracle_jacobi.restype = C.c_long
    L.oracle_residual.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double]
    L.oracle_residual.restype = C.c_double
    L.oracle_residual_ex.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.oracle_residual_ex.restype = C.c_double
    L.oracle_floodfill.argtypes = [_u32p, C.c_int, C.c_int]
    L.oracle_floodfill.restype = C.c_int
    L.oracle_fracts_3d.argtypes = [_dp, C.c_long, C.c_double, C.c_double, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]
    L.oracle_fracts_3d.restype = None
    if flavour is None:
        _LIB = L
    _LIBS[flavour] = L
    return L


OMEGA_REF = 2.0 / 3.0   # cuh:72


def synth_mask(nx, ny, seed=12345, img=0):
    pix = np.empty((ny, nx), dtype=np.uint8)
    lib().oracle_synth_mask(pix, nx, ny, seed, img)
    return pix


def porosity(pix):
    H, W = pix.shape
    return lib().oracle_porosity(np.ascontiguousarray(pix), W, H)


def fill_D_2phase(pix, DCF, DCS, ampX=1, ampY=1):
    H, W = pix.shape
    D = np.empty((H * ampY, W * ampX), dtype=np.float64)
    lib().oracle_fill_D_2phase(np.ascontiguousarray(pix), W, H, ampX, ampY, DCF, DCS, D)
    return D


def fill_D_3phase(pix, DCF, DCS, DCG, ampX=1, ampY=1):
    H, W = pix.shape
    D = np.empty((H * ampY, W * ampX), dtype=np.float64)
    lib().oracle_fill_D_3phase(np.ascontiguousarray(pix), W, H, ampX, ampY, DCF, DCS, DCG, D)
    return D


def linear_guess(nx, ny, CL, CR, flavour=None):
    x = np.empty((ny, nx), dtype=np.float64)
    lib(flavour).oracle_linear_guess(x, nx, ny, CL, CR)
    return x


def discretize(D, CL, CR, grid=None):
    ny, nx = D.shape
    A = np.empty((ny * nx, 5), dtype=np.float64)
    b = np.empty(ny * nx, dtype=np.float64)
    D = np.ascontiguousarray(D)
    if grid is None:
        lib().oracle_discretize_2d(D, A, b, nx, ny, 1.0 / nx, 1.0 / ny, CL, CR)
    else:
        g = np.ascontiguousarray(grid, dtype=np.uint32)
        lib().oracle_discretize_2d_impsolid(D, A, b, nx, ny, 1.0 / nx, 1.0 / ny, CL, CR, g)
    return A, b


def sweeps(A, b, x, nsweeps, kernel=0, omega=OMEGA_REF, flavour=None):
    ny, nx = x.shape
    x = np.array(x, dtype=np.float64, order="C", copy=True)
    tmp = np.empty_like(x)
    lib(flavour).oracle_sweeps(A, b, x, tmp, nx, ny, nsweeps, kernel, omega)
    return x


def flux_deff(x, D, CL, CR):
    ny, nx = x.shape
    MFL = np.empty(ny)
    MFR = np.empty(ny)
    d = lib().oracle_flux_deff(np.ascontiguousarray(x), np.ascontiguousarray(D), nx, ny, 1.0 / nx,
                               CL, CR, MFL, MFR)
    return d, MFL, MFR


def jacobi(A, b, x0, D, CL, CR, tol, max_iter, check_every=10000, kernel=0, omega=OMEGA_REF,
           flavour=None):
    """Returns (iters, deff_raw, conv, field, MFL, MFR)."""
    ny, nx = x0.shape
    x = np.array(x0, dtype=np.float64, order="C", copy=True)
    tmp = np.empty_like(x)
    MFL = np.zeros(ny)
    MFR = np.zeros(ny)
    deff = C.c_double(0)
    conv = C.c_double(0)
    it = lib(flavour).oracle_jacobi(A, b, x, tmp, nx, ny, CL, CR, tol, int(max_iter), int(check_every),
                             np.ascontiguousarray(D), MFL, MFR, kernel, omega,
                             C.byref(deff), C.byref(conv))
    return it, deff.value, conv.value, x, MFL, MFR


def residual(x, D, CL, CR, flavour=None):
    """Residual(), cuh:451-494: mean over the cells of |qW - qE + qN - qS| for field x and diffusivities D (both (ny, nx))."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    ny, nx = x.shape
    assert D.shape == x.shape
    return float(lib(flavour).oracle_residual(x, D, ny, nx, CL, CR))


def residual_exact(x, D, CL, CR):
    """(serial, exact): Residual() in the reference's order, and the same per-cell doubles added in long double (a yardstick
    for reductions in another order; see oracle_residual_ex)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    ny, nx = x.shape
    ex = C.c_double()
    r = lib().oracle_residual_ex(x, D, ny, nx, CL, CR, C.byref(ex))
    return float(r), ex.value


def assert_residual(got, x, D, CL, CR):
    """A residual reduced in another order than the reference's serial sum: within 1e-13 of the exactly added per-cell terms
    (tree sums lose ~log2(n) ulps), and within the serial sum's own error bound (n * 2^-53, at least 1e-12) of the oracle."""
    serial, exact = residual_exact(x, D, CL, CR)
    if not np.isfinite(serial):
        assert not np.isfinite(got)
        return
    assert abs(got - exact) <= 1e-13 * exact, (got, exact, serial)
    assert abs(got - serial) <= max(1e-12, x.size * 2.0 ** -53) * serial, (got, serial)


def floodfill(grid):
    """grid: (ny, nx) uint32 with 1 = solid; returns (grid with unreachable cells = 2, PathFlag)."""
    g = np.array(grid, dtype=np.uint32, order="C", copy=True)
    ny, nx = g.shape
    flag = lib().oracle_floodfill(g, nx, ny)
    return g, bool(flag)


def fracts_3d(D, DCS, DCF):
    s = C.c_double()
    l = C.c_double()
    D = np.ascontiguousarray(D)
    lib().oracle_fracts_3d(D, D.size, DCS, DCF, C.byref(s), C.byref(l))
    return s.value, l.value


def solve_3phase(pix, DCS, DCF, DCG, CL, CR, tol, max_iter, flavour=None, precond_max_iter=1000000):
    """SingleSim3Phase / BatchSim3Phase, cuh:1316-1633 (preCond = true, cuh:1443), on the oracle:
    Grid from pixels > 200 + FloodFill; gas diffusivity ramped 10, 100, ... < DCG with
    tolerance x10 and MAX_ITER 1e6 (JacobiGPUPreCond), each stage warm-started; then the real
    DCG with the user's tolerance (JacobiGPU).  `precond_max_iter` replaces the reference's literal 1e6
    (cuh:1503) so that a large image fits an oracle run (deff2d --precond-maxiter is the same knob).
    Returns dict(stage_sweeps, deff, conv, SVF, LVF, field, grid, path)."""
    ny, nx = pix.shape
    grid, path = floodfill((pix > 200).astype(np.uint32))
    x = linear_guess(nx, ny, CL, CR)
    stages = []
    g = 10.0
    while g < DCG:                                   # cuh:1492
        D = fill_D_3phase(pix, DCF, DCS, g)
        A, b = discretize(D, CL, CR, grid=grid)
        it, _, _, x, _, _ = jacobi(A, b, x, D, CL, CR, tol * 10, precond_max_iter, flavour=flavour)
        stages.append(it)
        g = g * 10
    D = fill_D_3phase(pix, DCF, DCS, DCG)
    svf, lvf = fracts_3d(D, DCS, DCF)
    A, b = discretize(D, CL, CR, grid=grid)
    it, deff, conv, x, _, _ = jacobi(A, b, x, D, CL, CR, tol, max_iter, flavour=flavour)
    stages.append(it)
    return dict(stage_sweeps=stages, deff=deff / DCF, conv=conv, SVF=svf, LVF=lvf, field=x, grid=grid, path=path)


def solve_single_2phase_ramp(pix, DCS, DCF_max, CL, CR, tol, max_iter, ampX=1, ampY=1, flavour=None):
    """SingleSim's DCF continuation, cuh:1713-1817, on the oracle: the fluid diffusivity is ramped
    DCF = 100^count (100, 1e4, 1e6, ...) clipped to Df; every stage refills D, re-assembles and runs
    JacobiGPU warm-started from the previous stage's field (x_vec is in/out, cuh:1793); after each
    stage `deff /= DCF` (cuh:1802).  The loop condition is tested on the PREVIOUS DCF, starting from
    10 (cuh:1714, :1761): for Df < 10 it never runs (the reference then prints uninitialised
    memory -- returned here as stages == []).
    Returns dict(stages=[(DCF, sweeps, deff_normalised, conv), ...], field)."""
    H, W = pix.shape
    nx, ny = W * ampX, H * ampY
    x = linear_guess(nx, ny, CL, CR, flavour=flavour)           # cuh:1730-1734
    DCF = 10.0
    count = 1
    stages = []
    while DCF <= DCF_max:                                        # cuh:1761
        DCF = float(100 ** count)                                # std::pow(100, count)
        if DCF >= DCF_max:
            DCF = DCF_max
        D = fill_D_2phase(pix, DCF, DCS, ampX, ampY)             # cuh:1773-1785
        A, b = discretize(D, CL, CR)                             # cuh:1789
        it, deff, conv, x, _, _ = jacobi(A, b, x, D, CL, CR, tol, max_iter, flavour=flavour)   # cuh:1793
        stages.append((DCF, it, deff / DCF, conv))               # cuh:1802
        if DCF == DCF_max:
            break
        count += 1
    return dict(stages=stages, field=x)


# ---- oracle/_ref/ref_host: the reference's OWN host-only functions (Deff2D.cuh with the CUDA-dependent lines cut out, see
# oracle/ref_host_probe.cpp and oracle/Makefile) as a checker.  Built where /root/reference is mounted; the binary travels to
# the GPU box with the snapshot. ----------------------------------------------------------------------------------------------
REF_HOST = os.path.join(ORACLE_DIR, "_ref", "ref_host")


def have_ref_host():
    return os.access(REF_HOST, os.X_OK)


def _ref_run(args, tmpdir, payload=None, out_bytes=None):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        if payload is not None:
            with open(fin, "wb") as f:
                f.write(payload)
        cmd = [REF_HOST] + [a.replace("@in", fin).replace("@out", fout) for a in args]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (cmd, r.stderr)
        data = open(fout, "rb").read() if out_bytes else None
        return r.stdout, data


def ref_discretize(D, CL, CR, grid=None, tmpdir=None):
    """The reference's DiscretizeMatrix2D / DiscretizeMatrix2D_ImpSolid (cuh:815-902 / cuh:715-812), its own code."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    ny, nx = D.shape
    payload = np.array([nx, ny, 0 if grid is None else 1], dtype=np.int32).tobytes() + np.array([CL, CR]).tobytes() + D.tobytes()
    if grid is not None:
        payload += np.ascontiguousarray(grid, dtype=np.uint32).tobytes()
    _, data = _ref_run(["assemble", "@in", "@out"], tmpdir, payload, True)
    a = np.frombuffer(data, dtype=np.float64)
    n = nx * ny
    return a[:5 * n].reshape(n, 5).copy(), a[5 * n:].copy()


def ref_residual(x, D, CL, CR, tmpdir=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    ny, nx = x.shape
    payload = np.array([nx, ny], dtype=np.int32).tobytes() + np.array([CL, CR]).tobytes() + x.tobytes() + np.ascontiguousarray(D, dtype=np.float64).tobytes()
    out, _ = _ref_run(["residual", "@in"], tmpdir, payload)
    return float(out)


def ref_floodfill(grid, tmpdir=None):
    g = np.ascontiguousarray(grid, dtype=np.uint32)
    ny, nx = g.shape
    _, data = _ref_run(["floodfill", "@in", "@out"], tmpdir, np.array([nx, ny], dtype=np.int32).tobytes() + g.tobytes(), True)
    return np.frombuffer(data, dtype=np.uint32).reshape(ny, nx).copy()


def ref_fractions(pix, D, DCS, DCF, tmpdir=None):
    """(porosity, SVF, LVF) by the reference's calcPorosity / calcFracts3D."""
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    ny, nx = pix.shape
    payload = (np.array([nx, ny], dtype=np.int32).tobytes() + np.array([DCS, DCF]).tobytes() + pix.tobytes()
               + np.ascontiguousarray(D, dtype=np.float64).tobytes())
    out, _ = _ref_run(["fractions", "@in"], tmpdir, payload)
    return tuple(float(v) for v in out.split())


def ref_fill_D(pix, DCS, DCF, DCG=0.0, ampX=1, ampY=1, phases=2, tmpdir=None):
    """The drivers' own D-fill loops (BatchSim cuh:1988-2000 / SingleSim3Phase cuh:1510-1531), mesh amplification included."""
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    H, W = pix.shape
    payload = np.array([W, H, ampX, ampY, phases], dtype=np.int32).tobytes() + np.array([DCS, DCF, DCG]).tobytes() + pix.tobytes()
    _, data = _ref_run(["fill", "@in", "@out"], tmpdir, payload, True)
    return np.frombuffer(data, dtype=np.float64).reshape(H * ampY, W * ampX).copy()


def ref_linear_guess(nx, ny, CL, CR, tmpdir=None):
    """The drivers' linear initial guess (cuh:1955-1959)."""
    payload = np.array([nx, ny], dtype=np.int32).tobytes() + np.array([CL, CR]).tobytes()
    _, data = _ref_run(["guess", "@in", "@out"], tmpdir, payload, True)
    return np.frombuffer(data, dtype=np.float64).reshape(ny, nx).copy()


def ref_whm(w1, w2, x1, x2):
    out, _ = _ref_run(["whm", repr(w1), repr(w2), repr(x1), repr(x2)], None)
    return float(out)


# ---- oracle/_ref/ref_kernel[_fma]: the reference's OWN sweep kernels (cuh:69-118), compiled by hipcc for gfx950 from their own
# text (oracle/ref_kernel_probe.hip, oracle/Makefile) -- needs a GPU to run. -------------------------------------------------
def have_ref_kernel():
    return os.access(os.path.join(ORACLE_DIR, "_ref", "ref_kernel"), os.X_OK)


def ref_sweeps(A, b, x, nsweeps, which=0, fma=False, tmpdir=None, timing=False):
    """nsweeps launches of the reference's updateX_SOR (which = 0) or updateX_V1 (1) on the GPU, with the reference's grid and its
    x <- xNew copy after every launch (cuh:1237-1281).  fma: the build with hipcc's default contraction."""
    import subprocess
    import tempfile
    x = np.ascontiguousarray(x, dtype=np.float64)
    ny, nx = x.shape
    exe = os.path.join(ORACLE_DIR, "_ref", "ref_kernel_fma" if fma else "ref_kernel")
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([nx, ny, int(nsweeps), int(which)], dtype=np.int32).tobytes())
            f.write(np.ascontiguousarray(A, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(b, dtype=np.float64).tobytes())
            f.write(x.tobytes())
        r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        out = np.fromfile(fout, dtype=np.float64).reshape(ny, nx)
        if timing:
            return out, float(r.stdout.split()[1])               # "loop_ms <ms> sweeps <n>"
        return out


# ---- oracle/_ref/ref_loop: the reference's host loop JacobiGPU (cuh:1163-1314) -- its own logic lines as fragments in a harness
# whose own lines move data through HIP (oracle/ref_loop_probe.hip) -- needs a GPU to run. ------------------------------------
def have_ref_loop():
    return os.access(os.path.join(ORACLE_DIR, "_ref", "ref_loop"), os.X_OK)


def ref_jacobi(A, b, x0, D, CL, CR, tol, max_iter, DCfluid=1.0, tmpdir=None, precond=False):
    """The reference's JacobiGPU (precond: JacobiGPUPreCond, which leaves deff / conv / gpu_ms untouched = 0) on the GPU:
    (iters, deff_raw, conv, field, MFL, MFR, gpu_ms).  The check interval is the reference's literal 10 000."""
    import struct
    import subprocess
    import tempfile
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    ny, nx = x0.shape
    n = nx * ny
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([nx, ny], dtype=np.int32).tobytes() + struct.pack("l", int(max_iter)) + np.array([tol, CL, CR, DCfluid]).tobytes())
            for a in (A, b, x0, D):
                f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        r = subprocess.run([os.path.join(ORACLE_DIR, "_ref", "ref_loop"), fin, fout] + (["precond"] if precond else []),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        raw = open(fout, "rb").read()
    iters = struct.unpack("l", raw[:8])[0]
    v = np.frombuffer(raw[8:], dtype=np.float64)
    return iters, float(v[0]), float(v[1]), v[3:3 + n].reshape(ny, nx).copy(), v[3 + n:3 + n + ny].copy(), v[3 + n + ny:3 + n + 2 * ny].copy(), float(v[2])
