"""The C++ mirror of the reference's seam (reference_seam.hpp): a driver written with
the reference's own function names and argument lists must compile, link and -- on
the GPU box -- reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def build_driver():
    # never rebuild libdeff_amd.so here: another test in this process may have it loaded
    if not os.path.exists(os.path.join(ROOT, "effectivediffusivityfvm_amd", "libdeff_amd.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "effectivediffusivityfvm_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", CPP], check=True)
    return os.path.join(CPP, "seam_driver")


def test_seam_driver_compiles_and_links():
    exe = build_driver()
    assert os.access(exe, os.X_OK)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_seam_driver_matches_oracle(oracle, img00000, tmp_path):
    exe = build_driver()
    raw = tmp_path / "pix.raw"
    img00000.tofile(raw)
    out = tmp_path / "out.bin"
    # BatchSim order with the reference's arrays; tol 1e-4 keeps it short (30 001 sweeps)
    r = subprocess.run([exe, str(raw), "128", "128", "1e-3", "1", "0", "1", "1e-4", "500000", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    data = np.fromfile(out, dtype=np.float64)
    n = 128 * 128
    iters, deff, conv = int(data[0]), data[1], data[2]
    x = data[4:4 + n].reshape(128, 128)
    A = data[4 + n:4 + 6 * n].reshape(n, 5)
    b = data[4 + 6 * n:4 + 7 * n]
    D = oracle.fill_D_2phase(img00000, 1.0, 1e-3)
    Ao, bo = oracle.discretize(D, 0.0, 1.0)
    it, d, c, xo, _, _ = oracle.jacobi(Ao, bo, oracle.linear_guess(128, 128, 0.0, 1.0), D, 0.0, 1.0, 1e-4, 500000)
    assert np.array_equal(A, Ao) and np.array_equal(b, bo)           # DiscretizeMatrix2D drop-in: bit-exact
    assert iters == it
    assert abs(deff - d / 1.0) <= 1e-8 * abs(d) and deff == d        # north_star 1e-8; in fact bit-exact
    assert conv == c
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-6 and np.array_equal(x, xo)
