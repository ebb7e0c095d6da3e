import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def recorded():
    with open(os.path.join(GOLDEN, "reference_recorded.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def img00000():
    """00000.jpg as decoded by the reference's own stb_image (tests/golden/make_stb_fixture.py)."""
    return np.load(os.path.join(GOLDEN, "img00000_pix_stb.npy"))


@pytest.fixture(scope="session")
def stb_recorded():
    with open(os.path.join(GOLDEN, "stb_decode_recorded.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small_cases():
    return dict(np.load(os.path.join(GOLDEN, "small_cases.npz")))


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    ob.build()
    return ob
