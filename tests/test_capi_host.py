"""Host-side checks that need no GPU: the C-ABI library loads, exports every
symbol include/deff_amd.h declares, and fails loudly (never falls back) when no
device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def capi():
    lib = os.path.join(ROOT, "effectivediffusivityfvm_amd", "libdeff_amd.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "effectivediffusivityfvm_amd", "csrc")], check=True)
    from effectivediffusivityfvm_amd import _capi
    _capi.load()
    return _capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "deff_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(deff_[a-z_0-9A-Z]+)\s*\(", text)))


def test_header_symbols_all_exported(capi):
    declared = _declared_symbols()
    assert len(declared) >= 20
    L = capi.load()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/deff_amd.h but not exported"
    assert sorted(capi.SYMBOLS) == declared


def test_version_and_error_strings(capi):
    L = capi.load()
    assert b"gfx950" in L.deff_version()
    assert L.deff_error_string(0) == b"ok"
    assert L.deff_error_string(-4) == b"no usable device"


def test_argument_validation_without_device(capi):
    L = capi.load()
    ctx = C.c_void_p()
    assert L.deff_create(0, 1, 8, C.byref(ctx)) == -1            # mesh too small: EINVAL before any HIP call
    assert b"2x2" in L.deff_last_error()
    assert L.deff_create(0, 8, 8, None) == -1
    assert L.deff_destroy(None) == 0
    assert L.deff_set_kernel(None, 0) == -1


def test_no_silent_fallback_without_gpu(capi):
    """On a machine without a GPU the product path must raise, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import effectivediffusivityfvm_amd as pkg
    with pytest.raises(pkg.DeffError) as ei:
        pkg.Solver(16, 16)
    assert ei.value.code == -4


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "effectivediffusivityfvm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_binding" not in text and "libdeff_oracle" not in text and "deff_oracle" not in text, f


def test_flood_fill_host_function_matches_oracle(capi, oracle, img00000, recorded):
    """deff_flood_fill is host code (no GPU): same Grid and PathFlag as the oracle's restatement
    of the reference's FloodFill, including the right-column seeding quirk."""
    import numpy as np
    import effectivediffusivityfvm_amd as pkg
    rng = np.random.default_rng(1)
    for _ in range(300):
        ny, nx = rng.integers(2, 16), rng.integers(2, 16)
        g = (rng.random((ny, nx)) < rng.uniform(0.1, 0.9)).astype(np.uint32)
        a, pa = oracle.floodfill(g)
        b, pb = pkg.flood_fill(g)
        assert np.array_equal(a, b) and pa == pb
    for thr in (150, 200):
        a, pa = oracle.floodfill((img00000 > thr).astype(np.uint32))
        b, pb = pkg.flood_fill((img00000 > thr).astype(np.uint32))
        assert np.array_equal(a, b) and pa == pb
    assert pkg.flood_fill((img00000 > 150).astype(np.uint32))[1] == bool(
        recorded["img00000_3phase_as_shipped"]["PathFlag_2phase"])
