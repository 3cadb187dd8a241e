"""One rank of a CPU rehearsal of the multi-GPU render, started by pathtracer_amd.launch.launch_ranks (or by
torch.distributed.run): reads RANK / WORLD_SIZE / MASTER_* from the environment, renders its row bands with the CPU
oracle in place of the GPU kernels, gathers over gloo with the same Gatherer bench.py uses, and prints one JSON line
on rank 0.  usage: multirank_child.py out.npz [fail_rank]"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

SX, SPP = 24, 3


def main():
    import pathtracer_amd as pt
    from harness import Oracle
    from pathtracer_amd import parallel

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if len(sys.argv) > 2 and int(sys.argv[2]) == rank:
        sys.exit(3)  # a rank that fails before the rendezvous
    dist.init_process_group("gloo", rank=rank, world_size=world)
    band = parallel.band_rows_for(SX, world)
    scene = pt.Scene.load(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    prm = pt.make_params(SX, 6, 7, 0.0, seed=9, rank=rank, n_ranks=world, band_rows=band)
    gat = parallel.Gatherer(SX, rank, world, band, torch.device("cpu"))
    for _ in range(2):  # two steps through the same buffers, as the benchmark does
        gat.reset()
        rgb_full, w_full, _ = Oracle(scene).render(prm, 0, SPP)  # touches only this rank's rows
        rows = pt.abi.global_rows(SX, rank, world, band)
        gat.rgb_local.copy_(torch.from_numpy(rgb_full.reshape(SX, SX, 3)[rows].copy()).reshape(-1))
        gat.wght_local.copy_(torch.from_numpy(w_full.reshape(SX, SX)[rows].copy()).reshape(-1))
        rgb, w = gat.gather()
    if rank == 0:
        np.savez(sys.argv[1], rgb=rgb.numpy(), w=w.numpy())
        print(json.dumps({"world": world, "band_rows": band, "rows": gat.counts}), flush=True)
    else:
        assert rgb is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
