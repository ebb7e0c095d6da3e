"""N > 1 path on CPU: two gloo ranks shard a batch of images, solve them with a
stand-in solver (the CPU oracle behind the Solver interface -- test infrastructure
only) and gather the NumImg x 9 table.  The table must equal the one-rank run
row for row, i.e. sharding and gathering change nothing."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT


class OracleSolver:
    """Same methods as effectivediffusivityfvm_amd.Solver, computed by the oracle (tests only)."""

    def __init__(self, ob):
        self.ob = ob

    def set_image(self, pix):
        self.pix = np.ascontiguousarray(pix)

    def assemble_2phase(self, Ds, Df, CL, CR):
        self.D = self.ob.fill_D_2phase(self.pix, Df, Ds)
        self.A, self.b = self.ob.discretize(self.D, CL, CR)
        self.CL, self.CR = CL, CR

    def init_linear(self, CL, CR):
        ny, nx = self.pix.shape
        self.x = self.ob.linear_guess(nx, ny, CL, CR)

    def solve(self, tol, max_iter, omega=2.0 / 3.0, check_every=10000):
        it, deff, conv, x, MFL, MFR = self.ob.jacobi(self.A, self.b, self.x, self.D, self.CL, self.CR, tol,
                                                     max_iter, check_every=check_every, omega=omega)
        self.x = x

        class R:
            pass
        r = R()
        r.iters, r.deff_raw, r.conv, r.loop_ms = it, deff, conv, 0.0
        return r


def _load(k):
    import oracle_binding as ob
    return ob.synth_mask(24, 16, 12345, k)


def _worker(rank, world, port, num_images, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_binding as ob
    from effectivediffusivityfvm_amd import batch
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    table = batch.run_batch(OracleSolver(ob), _load, num_images, 1e-3, 1.0, 0.0, 1.0, 1e-6, 200,
                            rank=rank, world=world, dist=dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "table.npy"), table)
    else:
        assert table is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("num_images", [5, 4, 1])
def test_two_rank_batch_equals_single_rank(oracle, tmp_path, num_images):
    import torch.multiprocessing as mp
    from effectivediffusivityfvm_amd import batch
    single = batch.run_batch(OracleSolver(oracle), _load, num_images, 1e-3, 1.0, 0.0, 1.0, 1e-6, 200)
    assert single.shape == (num_images, 9)
    assert list(single[:, 0]) == list(range(num_images))
    from conftest import spawn_with_timeout
    spawn_with_timeout(_worker, (2, _free_port(), num_images, str(tmp_path)), 2, timeout_s=240)
    table = np.load(tmp_path / "table.npy")
    assert np.array_equal(table, single)


def test_shard_is_a_partition():
    from effectivediffusivityfvm_amd import batch
    for n in (0, 1, 7, 8, 1024):
        for w in (1, 2, 4, 8):
            owned = [k for r in range(w) for k in batch.shard(n, r, w)]
            assert sorted(owned) == list(range(n))
            sizes = [len(batch.shard(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
