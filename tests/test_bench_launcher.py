"""`python bench.py --gpus N` must start its own N ranks: the parent stays a launcher that never
imports torch or touches the GPU, the ranks are fresh children of a `python -m torch.distributed.run` child -- the form the
driver itself uses for N > 1.  Proven here without GPUs: --launch-dry brings the ranks up on a gloo group; without it, on
a box with fewer GPUs than asked, the refusal comes from the ranks themselves, not from argument parsing."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("TORCHELASTIC_")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_line(text):
    for line in text.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text)


@pytest.mark.parametrize("mode", ["images", "slab"])
def test_gpus_2_starts_two_fresh_ranks(mode):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-dry", "--mode", mode], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _json_line(r.stdout)
    assert out["launch_dry"] and out["n_gpus"] == 2
    ranks = out["ranks"]
    assert [(x["rank"], x["world"], x["local_rank"]) for x in ranks] == [(0, 2, 0), (1, 2, 1)]
    assert len({x["pid"] for x in ranks}) == 2                 # two processes ...
    assert len({x["ppid"] for x in ranks}) == 1                # ... children of ONE launcher child, not of each other


def test_launch_dry_under_the_drivers_own_launcher():
    """The driver's form: torch.distributed.run around bench.py; the self-launch must stay out of the way."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--launch-dry"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _json_line(r.stdout)
    assert [x["rank"] for x in out["ranks"]] == [0, 1]


def test_gpus_1_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--launch-dry"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["ranks"][0]["rank"] == 0


def test_too_few_gpus_is_reported_by_the_ranks():
    import torch
    ndev = torch.cuda.device_count()
    if ndev >= 2:
        pytest.skip("this box has the GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    import re
    # (whichever rank gets to say it first: torchrun stops the other ranks as soon as one has failed)
    assert re.search(rf"bench\.py rank [01]: device [01] of {ndev} ", r.stderr), r.stderr
    assert "must be launched with" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode,n", [("images", 2), ("slab", 3)])
def test_rehearsal_of_n_ranks_sharing_one_gpu(mode, n):
    """The N > 1 paths of bench.py on hardware, as far as a one-GPU box allows: `python bench.py --gpus N --share-gpu --transport
    host` -- self-launched ranks, each with its own context (images) or its slab of one image (slab), barrier, max-over-ranks
    timing, one JSON line from rank 0 -- with the ranks sharing the GPU and gloo / host-staged halo blocks in place of RCCL
    (which refuses two ranks on one device).  What stays unexecuted is the RCCL transport itself."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--share-gpu", "--transport", "host", "--mode", mode, "--size", "1024",
                        "--steps", "2", "--warmup", "1", "--sweeps-per-step", "64"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == n and len(out["per_rank_ms_per_step"]) == n and out["steps"] == 2
    assert "REHEARSAL" in out["config"]["workload"]
    cells = 1024.0 * 1024.0 * (n if mode == "images" else 1)
    assert abs(out["value"] - cells * 64 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]
    assert out["scaling"] == ("weak" if mode == "images" else "strong")


@pytest.mark.gpu
def test_share_gpu_with_rccl_is_refused_by_the_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--steps", "1"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "--share-gpu needs --transport host" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["images", "slab"])
def test_one_rank_under_the_drivers_launcher_with_rccl(mode):
    """The one RCCL path a one-GPU box can execute: the driver's own form, `python -m torch.distributed.run --nproc-per-node 1
    ... bench.py --gpus 1`, brings up the 'nccl' (= RCCL) process group on the device, and the barrier and the gather of the
    ranks' times that bracket the timed region go through it."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "1", "--mode", mode, "--size", "1024", "--steps", "2",
                        "--warmup", "1", "--sweeps-per-step", "64", "--primary-only"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and len(out["per_rank_ms_per_step"]) == 1
    assert abs(out["value"] - 1024.0 * 1024.0 * 64 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]
