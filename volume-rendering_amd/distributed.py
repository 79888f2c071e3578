"""Multi-GPU frame split: one process per GPU, screen-space partition, one collective.

Rays are independent, so the frame shards naturally: every rank holds the whole volume / TF / ESL (replicated; 1 GiB
at 1024^3) and renders its own bands of rows; the only exchange is collecting the RGBA8 framebuffer on the display
rank — `torch.distributed.gather`, which on the "nccl" backend is RCCL send/recv over xGMI (each rank reaches the root
in one hop on the fully connected mesh; payload 16 MiB / world at 2048^2).  The reference has no multi-GPU code at all
(it picks one device, VolR.cpp:141-172); nothing here is translated from it.

The same code runs on the "gloo" backend with CPU tensors — that is how the partition / gather logic is tested
without GPUs (tests/test_distributed.py).
"""
import torch
import torch.distributed as dist

from .scene import band_partition


class FrameSplit:
    """Partition of an H x W frame over `world` ranks in interleaved bands of `band_rows` rows."""

    def __init__(self, width, height, world, rank, band_rows=None):
        if band_rows is None:
            band_rows = default_band_rows(height, world)
        self.width, self.height, self.world, self.rank, self.band_rows = width, height, world, rank, band_rows
        nbands = -(-height // band_rows)
        self.per_rank = -(-nbands // world)
        self.local_rows = self.per_rank * band_rows

    def apply(self, params):
        """Fills the partition fields of a vr_params for this rank."""
        p, per_rank = band_partition(params, self.rank, self.world, self.band_rows)
        assert per_rank == self.per_rank
        return p

    def local_buffer(self, device):
        return torch.empty((self.local_rows, self.width, 4), dtype=torch.uint8, device=device)

    def staging_buffer(self, device):
        return torch.empty((self.world, self.local_rows, self.width, 4), dtype=torch.uint8, device=device)

    def assemble(self, staging):
        """[world, per_rank*band_rows, W, 4] gathered buffers -> [H, W, 4] frame (band b of rank r is frame band b*world + r)."""
        w, pr, br = self.world, self.per_rank, self.band_rows
        if pr == 1:
            full = staging.reshape(w * br, self.width, 4)
        else:
            full = staging.reshape(w, pr, br, self.width, 4).permute(1, 0, 2, 3, 4).reshape(pr * w * br, self.width, 4)
        return full[: self.height]

    def gather(self, local, staging=None, dst=0, group=None):
        """Collects every rank's bands on `dst`; returns the assembled [H, W, 4] frame there, None elsewhere."""
        work, finish = self.gather_async(local, staging, dst, group)
        if work is not None:
            work.wait()
        return finish()

    def gather_async(self, local, staging=None, dst=0, group=None):
        """Starts the gather (it runs on the backend's own stream, ordered after the work already queued on the current
        stream) and returns (work, finish): `work.wait()` orders the current stream after the transfer, `finish()` then
        returns the assembled frame on `dst` (None elsewhere).  Lets frame i+1 render while frame i is collected."""
        if self.world == 1 and not dist.is_initialized():
            return None, lambda: local[: self.height]
        if self.rank == dst:
            if staging is None:
                staging = self.staging_buffer(local.device)
            work = dist.gather(local, gather_list=[staging[r] for r in range(self.world)], dst=dst, group=group, async_op=True)
            return work, lambda: self.assemble(staging)
        work = dist.gather(local, gather_list=None, dst=dst, group=group, async_op=True)
        return work, lambda: None


def default_band_rows(height, world):
    """Interleaved bands balance long centre rays against short edge rays.  128-row bands (8 workgroup tile rows — the
    kernel walks its tiles in 8x8-tile blocks for cache locality) when every rank still gets at least two of them, one
    from each half of the frame; otherwise 16-row bands (one tile row).  One rank gets the whole frame."""
    if world <= 1:
        return max(1, height)
    if height % 128 == 0 and height // 128 >= 2 * world:
        return 128
    return 16
