"""Random sequences of C-ABI calls on one context against a model made of the CPU oracle.

The parity tests elsewhere each exercise one flow (image -> assembly -> guess -> sweeps / solve -> read).  A context is a
state machine -- field ping-pong, frozen images of a stack, resident intervals and their restart copy, a row dictionary that
belongs to one assembly, plans that depend on tuning -- and the bug ADVICE r03 found by reading (an unchecked resident interval
outliving the field it ran under) lives BETWEEN calls.  Here every seed draws ~30 calls at random from: new image, assembly
(2-phase with changing Ds / Df / walls, 3 pixel classes, from a D plane), linear guess, a caller's field, sweeps (omega 2/3 and
1), solve (small MAX_ITER / check interval: stops by tolerance, by MAX_ITER on and between checks), wall fluxes, residual,
field read-back, and changes of kernel / sweeps per pass / tile form / launch mode in between.  After every call that returns
numbers they are compared with the oracle's: fields, sweep counts, Deff and conv bit for bit, the residual as
oracle_binding.assert_residual does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(64, 48), (130, 70), (97, 41), (256, 130), (40, 300), (512, 512)]


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


class Model:
    """The oracle's side of one stack of B images."""

    def __init__(self, ob, nx, ny, B, rng):
        self.ob, self.nx, self.ny, self.B, self.rng = ob, nx, ny, B, rng
        self.pix = None
        self.D = self.A = self.b = None
        self.x = None
        self.CL, self.CR = 0.0, 1.0
        self.kind = None                                     # "2p" / "3p" / "D": how the current system was assembled

    def new_image(self):
        p = self.rng.uniform(0.3, 0.7)
        vals = np.array([0, 30, 120, 199, 201, 255], dtype=np.uint8)
        if self.rng.random() < 0.5:
            self.pix = [np.where(self.rng.random((self.ny, self.nx)) < p, 0, 255).astype(np.uint8) for _ in range(self.B)]
        else:
            self.pix = [self.rng.choice(vals, size=(self.ny, self.nx)) for _ in range(self.B)]
        self.kind = None

    def assemble(self, kind, Ds, Df, Dg, CL, CR):
        ob = self.ob
        self.CL, self.CR, self.kind = CL, CR, kind
        if kind == "2p":
            self.D = [ob.fill_D_2phase(p, Df, Ds) for p in self.pix]
        elif kind == "3p":
            self.D = [ob.fill_D_3phase(p, Df, Ds, Dg) for p in self.pix]
        else:
            levels = np.array([Ds, Df, Dg, 0.5 * (Ds + Df)])
            self.D = [levels[self.rng.integers(0, 4, size=(self.ny, self.nx))] for _ in range(self.B)]
        sys_ = [ob.discretize(D, CL, CR) for D in self.D]
        self.A, self.b = [s[0] for s in sys_], [s[1] for s in sys_]

    def stacked(self, arrs):
        return np.concatenate(arrs, axis=0)


def run_sequence(pkg, ob, seed):
    rng = np.random.default_rng(seed)
    nx, ny = SHAPES[rng.integers(len(SHAPES))]
    B = int(rng.choice([1, 1, 3]))
    m = Model(ob, nx, ny, B, rng)
    log = []
    with pkg.Solver(nx, ny, nimg=B) as s:
        def fields_equal():
            got = s.get_field()
            for k in range(B):
                assert np.array_equal(got[k * ny:(k + 1) * ny], m.x[k]), (seed, log, "field of image %d" % k)

        m.new_image()
        s.set_image(np.stack(m.pix))
        for step in range(30):
            ops = ["image", "assemble", "tune"]
            if m.kind is not None:
                ops += ["init", "set_field"]
                if m.x is not None:
                    ops += ["sweeps", "sweeps", "solve", "solve", "flux", "get", "residual"]
            op = ops[rng.integers(len(ops))]
            if op == "image":
                m.new_image()
                s.set_image(np.stack(m.pix))
                log.append("image")
            elif op == "assemble":
                kind = ["2p", "2p", "3p", "D"][rng.integers(4)]
                Ds, Df, Dg = [(1e-3, 1.0, 10.0), (1e-2, 1.0, 50.0), (0.1, 2.0, 7.0)][rng.integers(3)]
                CL, CR = [(0.0, 1.0), (0.25, 0.75)][rng.integers(2)]
                m.assemble(kind, Ds, Df, Dg, CL, CR)
                if kind == "2p":
                    s.assemble_2phase(Ds, Df, CL, CR)
                elif kind == "3p":
                    s.assemble_3phase(Ds, Df, Dg, CL, CR)
                else:
                    s.set_kernel("auto")                     # a caller's D plane: explicit planes, dictionary if it has few rows
                    s.assemble_from_D(m.stacked(m.D), CL, CR)
                log.append(f"assemble {kind} {Ds} {Df} {CL}")
            elif op == "tune":
                what = rng.integers(5)
                if what == 0:
                    v = int(rng.choice([0, 2, 4, 8])); s.set_tuning("tb_T", v); log.append(f"tb_T {v}")
                elif what == 1:
                    v = int(rng.choice([0, 1, 2])); s.set_tuning("tb_impl", v); log.append(f"tb_impl {v}")
                elif what == 2:
                    v = int(rng.choice([0, 1])); s.set_tuning("tb_launch", v); log.append(f"tb_launch {v}")
                elif what == 3:
                    v = int(rng.choice([0, 1])); s.set_tuning("flux_reduce", v); log.append(f"flux_reduce {v}")
                else:
                    k = ["auto", "matfree_tb", "explicit", "matfree"][rng.integers(4)] if m.kind == "2p" else "auto"
                    s.set_kernel(k); log.append(f"kernel {k}")
            elif op == "init":
                s.init_linear(m.CL, m.CR)
                m.x = [ob.linear_guess(nx, ny, m.CL, m.CR) for _ in range(B)]
                log.append("init")
            elif op == "set_field":
                m.x = [rng.random((ny, nx)) for _ in range(B)]
                s.set_field(m.stacked(m.x))
                log.append("set_field")
            elif op == "sweeps":
                k = int(rng.choice([1, 3, 8, 17, 40, 64]))
                omega, kern = [(2.0 / 3.0, 0), (1.0, 1)][rng.integers(2)]
                s.sweeps(k, omega)
                m.x = [ob.sweeps(m.A[i], m.b[i], m.x[i], k, kernel=kern, omega=omega) for i in range(B)]
                log.append(f"sweeps {k} {omega:.3f}")
                fields_equal()
            elif op == "solve":
                tol = float(rng.choice([1e-2, 1e-4, 1e-12]))
                max_iter = int(rng.choice([1, 7, 100, 101, 130, 301]))
                ce = int(rng.choice([7, 50, 100]))
                res = s.solve(tol, max_iter, check_every=ce)
                res = [res] if B == 1 else res
                log.append(f"solve {tol} {max_iter} {ce}")
                for i in range(B):
                    it, deff, conv, x, MFL, MFR = ob.jacobi(m.A[i], m.b[i], m.x[i], m.D[i], m.CL, m.CR, tol, max_iter, check_every=ce)
                    assert (res[i].iters, res[i].deff_raw, res[i].conv) == (it, deff, conv), (seed, log, i)
                    m.x[i] = x
                fields_equal()
            elif op == "flux":
                d, MFL, MFR = s.flux()
                d = [d] if B == 1 else list(d)
                for i in range(B):
                    want, L, R = ob.flux_deff(m.x[i], m.D[i], m.CL, m.CR)
                    assert d[i] == want and np.array_equal(MFL[i * ny:(i + 1) * ny], L) and np.array_equal(MFR[i * ny:(i + 1) * ny], R), (seed, log, i)
                log.append("flux")
            elif op == "get":
                fields_equal()
                log.append("get")
            elif op == "residual":
                got = s.residual() if m.kind in ("2p", "3p") else s.residual(m.stacked(m.D), m.CL, m.CR)
                got = [got] if B == 1 else list(got)
                log.append("residual")
                for i in range(B):
                    try:
                        ob.assert_residual(got[i], m.x[i], m.D[i], m.CL, m.CR)
                    except AssertionError as e:
                        raise AssertionError((seed, log, i, e.args))
        if m.x is not None:
            fields_equal()
        assert s.plan_value("tb_fallbacks") == 0
    return log


@pytest.mark.parametrize("seed", range(40))
def test_random_call_sequences_match_the_oracle(pkg, oracle, seed):
    run_sequence(pkg, oracle, 1000 + seed)


def run_slab_sequence(pkg, ob, seed):
    """The same idea for one image split into row slabs (deff_slab_group_*, N slabs on one GPU: peer copies between their
    buffers, one exchange of 8 halo rows per pass, optionally overlapped with the interior): random sizes (odd widths, slabs
    that differ by a row), slab counts, sweeps per pass, overlap modes and call sequences; every read against the oracle."""
    rng = np.random.default_rng(seed)
    nx = int(rng.integers(8, 420))
    nslabs = int(rng.integers(2, 5))
    ny = int(rng.integers(8 * nslabs, 8 * nslabs + 500))
    m = Model(ob, nx, ny, 1, rng)
    log = [f"nx {nx} ny {ny} slabs {nslabs}"]
    with pkg.SlabGroup(nx, ny, [0] * nslabs) as g:
        m.new_image()
        g.set_image(m.pix[0])
        for step in range(16):
            ops = ["image", "assemble", "tune"]
            if m.kind is not None:
                ops += ["init", "set_field"]
                if m.x is not None:
                    ops += ["sweeps", "sweeps", "solve", "flux", "get"]
            op = ops[rng.integers(len(ops))]
            if op == "image":
                m.new_image()
                g.set_image(m.pix[0])
                log.append("image")
            elif op == "assemble":
                kind = ["2p", "2p", "3p"][rng.integers(3)]
                Ds, Df, Dg = [(1e-3, 1.0, 10.0), (1e-2, 1.0, 50.0), (0.1, 2.0, 7.0)][rng.integers(3)]
                CL, CR = [(0.0, 1.0), (0.25, 0.75)][rng.integers(2)]
                m.assemble(kind, Ds, Df, Dg, CL, CR)
                if kind == "2p":
                    g.assemble_2phase(Ds, Df, CL, CR)
                else:
                    g.assemble_3phase(Ds, Df, Dg, CL, CR)
                log.append(f"assemble {kind} {Ds} {Df} {CL}")
            elif op == "tune":
                if rng.random() < 0.5:
                    v = int(rng.choice([0, 1, 2, 4, 8])); g.set_tuning("tb_T", v); log.append(f"tb_T {v}")
                else:
                    v = int(rng.choice([0, 1, 2])); g.set_tuning("slab_overlap", v); log.append(f"slab_overlap {v}")
            elif op == "init":
                g.init_linear(m.CL, m.CR)
                m.x = [ob.linear_guess(nx, ny, m.CL, m.CR)]
                log.append("init")
            elif op == "set_field":
                m.x = [rng.random((ny, nx))]
                g.set_field(m.x[0])
                log.append("set_field")
            elif op == "sweeps":
                k = int(rng.choice([1, 3, 8, 17, 40]))
                omega, kern = [(2.0 / 3.0, 0), (1.0, 1)][rng.integers(2)]
                g.sweeps(k, omega)
                m.x = [ob.sweeps(m.A[0], m.b[0], m.x[0], k, kernel=kern, omega=omega)]
                log.append(f"sweeps {k} {omega:.3f}")
                assert np.array_equal(g.get_field(), m.x[0]), (seed, log)
            elif op == "solve":
                tol = float(rng.choice([1e-2, 1e-4, 1e-12]))
                max_iter = int(rng.choice([1, 7, 100, 101, 130]))
                ce = int(rng.choice([7, 50, 100]))
                r = g.solve(tol, max_iter, check_every=ce)
                log.append(f"solve {tol} {max_iter} {ce}")
                it, deff, conv, x, MFL, MFR = ob.jacobi(m.A[0], m.b[0], m.x[0], m.D[0], m.CL, m.CR, tol, max_iter, check_every=ce)
                assert (r.iters, r.deff_raw, r.conv) == (it, deff, conv), (seed, log)
                m.x = [x]
                assert np.array_equal(g.get_field(), x), (seed, log)
            elif op == "flux":
                d, MFL, MFR = g.flux()
                want, L, R = ob.flux_deff(m.x[0], m.D[0], m.CL, m.CR)
                assert d == want and np.array_equal(MFL, L) and np.array_equal(MFR, R), (seed, log)
                log.append("flux")
            elif op == "get":
                assert np.array_equal(g.get_field(), m.x[0]), (seed, log)
                log.append("get")
    return log


@pytest.mark.parametrize("seed", range(16))
def test_random_call_sequences_on_row_slabs_match_the_oracle(pkg, oracle, seed):
    run_slab_sequence(pkg, oracle, 5000 + seed)
