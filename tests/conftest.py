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


def spawn_with_timeout(fn, args, nprocs, timeout_s=300):
    """torch.multiprocessing.spawn with a deadline: fresh child processes (spawn start method), joined in a loop; when the
    deadline passes the CHILDREN are killed (exact PIDs) and the test fails -- a wedged rank (a collective that never
    completes) must not stall the whole suite.  Nothing is re-executed and no process that touched the GPU is replaced."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.monotonic() + timeout_s
    try:
        while not ctx.join(timeout=5):
            if time.monotonic() > deadline:
                for p in ctx.processes:
                    if p.is_alive():
                        p.kill()
                for p in ctx.processes:
                    p.join(10)
                pytest.fail(f"{fn.__name__}: {nprocs} ranks did not finish within {timeout_s} s (children killed)")
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
