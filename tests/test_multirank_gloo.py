"""The multi-rank path on CPUs: two processes (gloo) split the rows in bands, each renders its share, rank 0 gathers.

The GPU kernels are replaced by the CPU oracle here (this is a test of the split and the exchange, which are the
same code for RCCL and gloo); the result must equal a single-process render bit for bit.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SX, SPP, BAND = 24, 3, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import pathtracer_amd as pt
    from harness import Oracle
    from pathtracer_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    prm = pt.make_params(SX, 6, 7, 0.0, seed=9, rank=rank, n_ranks=world, band_rows=BAND)
    rgb_full, w_full, _ = Oracle(scene).render(prm, 0, SPP)  # touches only this rank's rows
    rows = pt.abi.global_rows(SX, rank, world, BAND)
    assert len(rows) == parallel.local_row_count(SX, rank, world, BAND)
    rgb_local = torch.from_numpy(rgb_full.reshape(SX, SX, 3)[rows].copy()).reshape(-1)
    w_local = torch.from_numpy(w_full.reshape(SX, SX)[rows].copy()).reshape(-1)
    rgb, w = parallel.gather_accumulators(rgb_local, w_local, SX, rank, world, BAND)
    if rank == 0:
        np.savez(out_path, rgb=rgb.numpy(), w=w.numpy())
    else:
        assert rgb is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import pathtracer_amd as pt
    from harness import Oracle

    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    scene = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    rgb, w, _ = Oracle(scene).render(pt.make_params(SX, 6, 7, 0.0, seed=9), 0, SPP)
    assert np.array_equal(z["rgb"].reshape(-1), rgb) and np.array_equal(z["w"].reshape(-1), w)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_started_without_a_launcher(world, tmp_path):
    """pathtracer_amd.launch.launch_ranks, which `python bench.py --gpus N` uses when it was not started by
    torch.distributed.run: the parent starts the ranks, hands on rank 0's line, and the gathered image equals the
    single-process render bit for bit (uneven shares at three ranks: 24 rows in bands of 8)."""
    import json

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import pathtracer_amd as pt
    from harness import Oracle
    from pathtracer_amd import launch

    out = str(tmp_path / "gathered.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc, text = launch.launch_ranks([sys.executable, os.path.join(REPO, "tests", "multirank_child.py"), out], world, env=env, timeout=300)
    assert rc == 0, text
    line = json.loads(text.strip().splitlines()[-1])
    assert line["world"] == world and sum(line["rows"]) == SX
    z = np.load(out)
    scene = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    rgb, w, _ = Oracle(scene).render(pt.make_params(SX, 6, 7, 0.0, seed=9), 0, SPP)
    assert np.array_equal(z["rgb"].reshape(-1), rgb) and np.array_equal(z["w"].reshape(-1), w)


def test_a_failing_rank_fails_the_launch(tmp_path):
    from pathtracer_amd import launch

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GLOO_SOCKET_IFNAME"] = env.get("GLOO_SOCKET_IFNAME", "lo")
    rc, _ = launch.launch_ranks([sys.executable, os.path.join(REPO, "tests", "multirank_child.py"), str(tmp_path / "x.npz"), "0"], 1,
                                env=env, timeout=120)
    assert rc == 3


@pytest.mark.parametrize("world,bad", [(2, 1), (3, 2), (2, 0)])
def test_a_rank_that_dies_before_the_rendezvous_ends_the_launch_at_once(world, bad, tmp_path):
    """Rank `bad` exits with code 3 before init_process_group; the others are by then waiting for it at the
    rendezvous.  The launch must end them and return that code within seconds, not after the store's timeout."""
    import time

    from pathtracer_amd import launch

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GLOO_SOCKET_IFNAME"] = env.get("GLOO_SOCKET_IFNAME", "lo")
    t0 = time.monotonic()
    rc, _ = launch.launch_ranks([sys.executable, os.path.join(REPO, "tests", "multirank_child.py"), str(tmp_path / "x.npz"), str(bad)],
                                world, env=env, timeout=240)
    dt = time.monotonic() - t0
    assert rc == 3
    assert dt < 60, f"the launch took {dt:.0f} s to notice the dead rank"


def test_a_launch_that_runs_out_of_time_is_ended(tmp_path):
    import time

    from pathtracer_amd import launch

    t0 = time.monotonic()
    rc, out = launch.launch_ranks([sys.executable, "-c", "import time; print('started', flush=True); time.sleep(600)"], 2, timeout=3)
    assert rc == 124 and time.monotonic() - t0 < 30
    assert "started" in out  # what rank 0 had printed is still handed on


def test_band_height_shrinks_for_small_images():
    from pathtracer_amd import parallel

    assert parallel.band_rows_for(1024, 8) == 8
    assert parallel.band_rows_for(16, 8) == 2
    assert parallel.band_rows_for(4, 8) == 1
