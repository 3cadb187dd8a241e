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


def hip_runtime_copies():
    """Paths of the HIP runtime libraries mapped into this process.  A hipStream_t is only meaningful inside the
    copy of the runtime that created it: handing PyTorch's stream to libptx_hip.so (Device.set_stream) is valid
    when both resolved to one libamdhip64 (they do when the sonames agree: the dynamic linker loads one)."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


class Gatherer:
    """Accumulators of one rank and their gather to `dst`, with every buffer allocated once.

    `local` is one tensor: this rank's rgb rows (rows*sx*3 doubles) at the front, its weights (rows*sx doubles) at
    offset max_rows*sx*3, max_rows being the largest share of any rank, so that the whole tensor is the message
    and nothing is packed before a gather.  On `dst`, `rgb` and `wght` are the whole image (row order of the
    reference's rgbIm and wght, PathTracer.c:550-557)."""

    def __init__(self, sx, rank, world, band_rows, device, dst=0, group=None):
        self.sx, self.rank, self.world, self.band_rows, self.dst, self.group = sx, rank, world, band_rows, dst, group
        self.counts = [abi.rows_owned(sx, r, world, band_rows) for r in range(world)]
        self.rows = self.counts[rank]
        self.max_rows = max(self.counts)
        self.local = torch.zeros(self.max_rows * sx * 4, dtype=torch.float64, device=device)
        self.rgb_local = self.local[: self.rows * sx * 3]
        self.wght_local = self.local[self.max_rows * sx * 3: self.max_rows * sx * 3 + self.rows * sx]
        self.is_dst = rank == dst
        self.via_host = world > 1 and self.local.is_cuda and dist.get_backend(group) == "gloo"
        self.rgb = self.wght = None
        self.bufs = None
        if world == 1:
            self.rgb, self.wght = self.rgb_local.view(sx, sx, 3), self.wght_local.view(sx, sx)
        elif self.is_dst:
            self.rgb = torch.empty(sx, sx, 3, dtype=torch.float64, device=device)
            self.wght = torch.empty(sx, sx, dtype=torch.float64, device=device)
            buf_dev = "cpu" if self.via_host else device
            self.bufs = [torch.empty(self.max_rows * sx * 4, dtype=torch.float64, device=buf_dev) for _ in range(world)]
            self.row_index = [torch.as_tensor(abi.global_rows(sx, r, world, band_rows), device=device) for r in range(world)]
        if self.via_host:
            self.staged = torch.empty(self.max_rows * sx * 4, dtype=torch.float64, device="cpu")

    def reset(self):
        """rgb = 0, weights = 1 (newImage and the wght loop, PathTracer.c:550-557)."""
        self.rgb_local.zero_()
        self.wght_local.fill_(1.0)

    def gather(self):
        """Collects every rank's rows on dst; returns (rgb, wght) whole-image tensors there, (None, None) elsewhere."""
        sx = self.sx
        if self.world == 1:
            return self.rgb, self.wght
        if self.via_host:  # gloo moves host tensors only (a rehearsal of the GPU path without RCCL)
            self.staged.copy_(self.local)
            dist.gather(self.staged, self.bufs, dst=self.dst, group=self.group)
        else:
            dist.gather(self.local, self.bufs, dst=self.dst, group=self.group)
        if not self.is_dst:
            return None, None
        for r in range(self.world):
            n = self.counts[r]
            buf = self.bufs[r].to(self.rgb.device, non_blocking=True) if self.via_host else self.bufs[r]
            self.rgb.index_copy_(0, self.row_index[r], buf[: n * sx * 3].view(n, sx, 3))
            self.wght.index_copy_(0, self.row_index[r], buf[self.max_rows * sx * 3: self.max_rows * sx * 3 + n * sx].view(n, sx))
        return self.rgb, self.wght


def gather_accumulators(rgb_local, wght_local, sx, rank, world, band_rows, dst=0, group=None):
    """One-off form of Gatherer.gather for accumulators held elsewhere: copies them in, gathers, returns whole-image
    tensors on `dst` (None elsewhere).

    rgb_local: (rows_owned*sx*3,) float64, wght_local: (rows_owned*sx,) float64, in local row order.
    """
    if world == 1:
        return rgb_local.view(sx, sx, 3), wght_local.view(sx, sx)
    g = Gatherer(sx, rank, world, band_rows, rgb_local.device, dst=dst, group=group)
    g.rgb_local.copy_(rgb_local)
    g.wght_local.copy_(wght_local)
    return g.gather()
