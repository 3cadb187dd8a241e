"""Multi-GPU split of one render: one process per GPU, rows dealt out in bands.

Pixels are independent in the reference's sample loop (PathTracer.c:620-654)
while the samples of one pixel are not (the accumulator at :646-652 is an
ordered recurrence), so the image is split by rows and every rank runs all
samples of its own pixels.  Bands of `band_rows` rows go round-robin to the
ranks, which evens out the cost difference between image regions.  The only
exchange is the gather of the finished accumulators to one rank, which the
host needs before it can write the image (dataOutput, PathTracer.c:655,
:661).  torch.distributed supplies the transport: RCCL over xGMI with the
"nccl" backend on GPUs, gloo on CPUs.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import abi


def band_rows_for(sx, world):
    """Band height: 8 rows (one wave tile) unless the image is too small to give every rank a band."""
    b = 8
    while b > 1 and (sx + b - 1) // b < world:
        b //= 2
    return b


def local_row_count(sx, rank, world, band_rows):
    return abi.rows_owned(sx, rank, world, band_rows)


_ROW_INDEX_CACHE = {}


def _global_rows_tensor(sx, r, world, band_rows, device):
    """Whole-image row numbers of rank r's rows as an index tensor on `device` (built once per split)."""
    key = (sx, r, world, band_rows, str(device))
    t = _ROW_INDEX_CACHE.get(key)
    if t is None:
        t = torch.as_tensor(abi.global_rows(sx, r, world, band_rows), device=device)
        _ROW_INDEX_CACHE[key] = t
    return t


def gather_accumulators(rgb_local, wght_local, sx, rank, world, band_rows, dst=0, group=None):
    """Collects every rank's rows on `dst` and returns whole-image tensors there (None elsewhere).

    rgb_local: (rows_owned*sx*3,) float64, wght_local: (rows_owned*sx,) float64, in local row order.
    """
    if world == 1:
        return rgb_local.view(sx, sx, 3), wght_local.view(sx, sx)
    counts = [abi.rows_owned(sx, r, world, band_rows) for r in range(world)]
    max_rows = max(counts)
    # one message per rank: rgb and wght of a row side by side, padded to the largest share
    packed = torch.zeros(max_rows, sx, 4, dtype=torch.float64, device=rgb_local.device)
    n = counts[rank]
    packed[:n, :, :3] = rgb_local.view(n, sx, 3)
    packed[:n, :, 3] = wght_local.view(n, sx)
    if packed.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only (a rehearsal of the GPU path without RCCL)
        staged = packed.cpu()
        host_bufs = [torch.empty_like(staged) for _ in range(world)] if rank == dst else None
        dist.gather(staged, host_bufs, dst=dst, group=group)
        bufs = [b.to(packed.device) for b in host_bufs] if rank == dst else None
    else:
        bufs = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
        dist.gather(packed, bufs, dst=dst, group=group)
    if rank != dst:
        return None, None
    rgb = torch.empty(sx, sx, 3, dtype=torch.float64, device=rgb_local.device)
    wght = torch.empty(sx, sx, dtype=torch.float64, device=rgb_local.device)
    for r in range(world):
        rows = _global_rows_tensor(sx, r, world, band_rows, rgb_local.device)
        rgb[rows] = bufs[r][: counts[r], :, :3]
        wght[rows] = bufs[r][: counts[r], :, 3]
    return rgb, wght
