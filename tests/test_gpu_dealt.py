"""Dealt tiles of the streaming kernel (round 4): chunk heights follow the order in which a SIMD serves its waves
(deal_ranked_tiles, api_solve.hip) -- a table of (strip, first row, rows) per wave replaces the equal chunks.  Whatever the
table says, every row must be swept exactly once: the results are the oracle's and those of equal chunks, bit for bit, for
ragged shapes, both pass lengths, skewed weights, many and few chunks per strip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import effectivediffusivityfvm_amd as p
    return p


def rand_mask(rng, nx, ny, p=0.5):
    return np.where(rng.random((ny, nx)) < p, 0, 255).astype(np.uint8)


def run(pkg, pix, x0, sweeps, tune, omega=2.0 / 3.0):
    ny, nx = pix.shape
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 1)
        for k, v in tune.items():
            s.set_tuning(k, v)
        s.set_image(pix)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.set_field(x0)
        s.sweeps(sweeps, omega)
        return s.get_field(), s.plan()


@pytest.mark.parametrize("T", [8, 6])
@pytest.mark.parametrize("shape", [(300, 200), (1030, 137), (250, 333), (2, 64), (97, 241), (1281, 410), (2050, 700), (131, 2000)])
def test_dealt_tiles_vs_oracle_and_equal_chunks(pkg, oracle, shape, T):
    nx, ny = shape
    rng = np.random.default_rng(nx * 7 + ny * 3 + T)
    pix = rand_mask(rng, nx, ny, 0.5)
    D = oracle.fill_D_2phase(pix, 1.0, 1e-3)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = rng.random((ny, nx))
    k = 3 * T + 3
    want = oracle.sweeps(A, b, x0, k)
    got, plan = run(pkg, pix, x0, k, {"tb_T": T})
    assert plan["tb_impl"] == 1 and plan["tb_T"] == T
    assert plan["tb_ranked"] == (1 if ny >= 3 * T and T == 8 else 0), plan   # passes of eight sweeps; at least one chunk of T rows per rank
    assert np.array_equal(got, want)
    flat, plan0 = run(pkg, pix, x0, k, {"tb_T": T, "tb_ranked": 0})
    assert plan0["tb_ranked"] == 0
    assert np.array_equal(flat, want)
    # skewed speeds and a heavy wall surcharge: another table, the same rows
    for tune in ({"tb_rank_w0": 900, "tb_rank_w1": 60, "tb_rank_w2": 40}, {"tb_rank_w0": 100, "tb_rank_w1": 300, "tb_rank_w2": 600},
                 {"tb_rank_wall": 2500}):
        skew, _ = run(pkg, pix, x0, k, dict(tune, tb_T=T))
        assert np.array_equal(skew, want), tune


def test_dealt_tiles_are_the_default_at_4096_and_for_stacks(pkg):
    with pkg.Solver(4096, 4096) as s:
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        p = s.plan()
        assert (p["tb_impl"], p["tb_T"], p["tb_ranked"], p["tb_blocks"]) == (1, 8, 1, 768), p
    with pkg.Solver(2048, 2048, nimg=4) as s:
        for k in range(4):
            s.synth_image(k + 1, k)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        s.sweeps(16)
        p = s.plan()
        assert p["tb_impl"] == 1 and p["tb_ranked"] == 1, p


def test_dealt_tiles_on_a_stack_whose_images_stop_at_different_checks(pkg, oracle):
    """Every (image, strip) column of a stack is cut into its own chunks; images that have met the tolerance leave the launch
    (their tiles return at once) and keep their field."""
    nx, ny, B = 300, 260, 5
    rng = np.random.default_rng(21)
    pixs = [rand_mask(rng, nx, ny, 0.35 + 0.1 * k) for k in range(B)]
    for ranked in (1, 0):
        with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
            s.set_tuning("tb_impl", 1); s.set_tuning("tb_T", 8); s.set_tuning("tb_ranked", ranked)
            s.set_image(np.stack(pixs))
            s.assemble_2phase(1e-2, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            s.sweeps(24)
            assert s.plan()["tb_ranked"] == ranked and s.plan()["tb_impl"] == 1
            s.init_linear(0.0, 1.0)
            res = s.solve(1e-3, 6000, check_every=200)
            got = s.get_field()
        iters = set()
        for k in range(B):
            D = oracle.fill_D_2phase(pixs[k], 1.0, 1e-2)
            A, b = oracle.discretize(D, 0.0, 1.0)
            it, deff, conv, x, _, _ = oracle.jacobi(A, b, oracle.linear_guess(nx, ny, 0.0, 1.0), D, 0.0, 1.0, 1e-3, 6000, check_every=200)
            assert (res[k].iters, res[k].deff_raw, res[k].conv) == (it, deff, conv)
            assert np.array_equal(got[k * ny:(k + 1) * ny], x)
            iters.add(it)
        assert len(iters) > 1


def test_dealt_tiles_on_a_dictionary_system_and_omega_one(pkg, oracle):
    """Three pixel classes with impermeable solid (the guarded kernel, b looked up in every strip) and with a permeable one (the
    unguarded kernel on dictionary rows: dealt tiles); and plain Jacobi."""
    nx, ny = 700, 500
    rng = np.random.default_rng(5)
    pix = rng.choice(np.array([0, 30, 120, 199, 201, 255], dtype=np.uint8), size=(ny, nx), p=[0.25, 0.1, 0.25, 0.1, 0.1, 0.2])
    pix[0] = pix[-1] = 255
    grid, _ = oracle.floodfill((pix > 200).astype(np.uint32))
    D = oracle.fill_D_3phase(pix, 1.0, 0.0, 50.0)
    with np.errstate(all="ignore"):
        A, b = oracle.discretize(D, 0.0, 1.0, grid=grid)
        x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
        for omega, kern in ((2.0 / 3.0, 0), (1.0, 1)):
            want = oracle.sweeps(A, b, x0, 35, kernel=kern, omega=omega)
            with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
                s.set_tuning("tb_impl", 1)
                s.set_image(pix)
                s.assemble_3phase(0.0, 1.0, 50.0, 0.0, 1.0, grid)
                s.init_linear(0.0, 1.0)
                s.sweeps(35, omega)
                assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_impl"] == 1
                print("guarded kernel, dealt:", s.plan()["tb_ranked"])
                assert np.array_equal(s.get_field(), want, equal_nan=True)


def test_dealt_tiles_on_three_permeable_classes(pkg, oracle):
    """Rows from the harvested dictionary (DiscretizeMatrix2D on a three-valued D plane), unguarded kernel: dealt tiles."""
    nx, ny = 700, 500
    rng = np.random.default_rng(6)
    pix = np.where(rng.random((ny, nx)) < 0.3, 255, np.where(rng.random((ny, nx)) < 0.5, 120, 0)).astype(np.uint8)
    pix[0] = pix[-1] = 255                                             # (keeps the distinct matrix rows under the dictionary's 511)
    D = oracle.fill_D_3phase(pix, 1.0, 0.5, 30.0)
    A, b = oracle.discretize(D, 0.0, 1.0)
    x0 = oracle.linear_guess(nx, ny, 0.0, 1.0)
    want = oracle.sweeps(A, b, x0, 35)
    with pkg.Solver(nx, ny, kernel="matfree_tb") as s:
        s.set_tuning("tb_impl", 1); s.set_tuning("tb_T", 8)
        s.set_image(pix)
        s.assemble_3phase(0.5, 1.0, 30.0, 0.0, 1.0, None)
        s.init_linear(0.0, 1.0)
        s.sweeps(35)
        assert s.kernel_in_use() == "matfree_tb" and s.plan()["tb_ranked"] == 1, s.plan()
        assert np.array_equal(s.get_field(), want)


def test_the_deal_watches_its_assumption(pkg):
    """On a GPU this process has to itself the waves land in the slots their tiles were cut for (no misses, dealing stays on);
    with a second context sweeping on another stream at the same time they do not, the count says so, and whichever way the
    watch decides the fields are the same bits."""
    import threading
    n = 4096
    with pkg.Solver(n, n) as s:
        s.synth_image(12345, 0)
        s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
        s.init_linear(0.0, 1.0)
        for _ in range(4):
            s.sweeps(64)
        assert s.plan()["tb_ranked"] == 1
        assert s.plan_value("tb_rank_lost") == 0 and s.plan_value("tb_rank_misses") <= 3072 * 8 // 50, s.plan_value("tb_rank_misses")
        alone = s.get_field()
    stop = threading.Event()

    def disturb():
        with pkg.Solver(2048, 2048, kernel="explicit") as d:
            d.synth_image(7, 0)
            d.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            d.init_linear(0.0, 1.0)
            while not stop.is_set():
                d.sweeps(50)

    th = threading.Thread(target=disturb)
    th.start()
    try:
        with pkg.Solver(n, n) as s:
            s.synth_image(12345, 0)
            s.assemble_2phase(1e-3, 1.0, 0.0, 1.0)
            s.init_linear(0.0, 1.0)
            for _ in range(4):
                s.sweeps(64)
            print("with a second context sweeping: misses", s.plan_value("tb_rank_misses"), "lost", s.plan_value("tb_rank_lost"),
                  "dealt in the last plan", s.plan()["tb_ranked"])
            shared = s.get_field()
    finally:
        stop.set()
        th.join()
    assert np.array_equal(alone, shared)


def test_dealt_tiles_fuzz_against_equal_chunks(pkg):
    """40 random contexts -- shape, stack size, weights, wall surcharge -- swept with dealt tiles and with equal chunks: every
    table must cover every row of every image exactly once, i.e. give the same field bit for bit."""
    rng = np.random.default_rng(2026)
    dealt = 0
    for case in range(40):
        nx = int(rng.integers(2, 1400))
        ny = int(rng.integers(24, 700))
        B = int(rng.choice([1, 1, 2, 3, 5]))
        if nx * ny * B > 1_500_000:
            B = 1
        pix = np.where(rng.random((B * ny, nx)) < 0.5, 0, 255).astype(np.uint8)
        x0 = rng.random((B * ny, nx))
        w = sorted((int(v) for v in rng.integers(50, 900, size=3)), reverse=True)
        tune = {"tb_rank_w0": w[0], "tb_rank_w1": w[1], "tb_rank_w2": w[2], "tb_rank_wall": int(rng.integers(1000, 2000))}
        k = int(rng.integers(8, 30))
        out = []
        for ranked in (1, 0):
            with pkg.Solver(nx, ny, nimg=B, kernel="matfree_tb") as s:
                s.set_tuning("tb_impl", 1); s.set_tuning("tb_T", 8); s.set_tuning("tb_ranked", ranked)
                for key, v in tune.items():
                    s.set_tuning(key, v)
                s.set_image(pix)
                s.assemble_2phase(1e-3, 1.0, 0.25, 0.75)
                s.set_field(x0)
                s.sweeps(k)
                out.append(s.get_field())
                if ranked:
                    dealt += s.plan()["tb_ranked"]
        assert np.array_equal(out[0], out[1]), (case, nx, ny, B, tune, k)
    assert dealt >= 30
