"""Multi-GPU split of a render: one process per GPU, samples-per-pixel sharded by sample index, one sum-reduce
of the float film over RCCL (torch.distributed backend "nccl" on ROCm) at the end.

The scene is replicated (it is read-only and small next to 288 GB of HBM); camera sample k of a pixel uses the
random stream keyed by (pixel, k) no matter which rank renders it, so the reduced film equals a single-GPU
render of all samples up to float summation order. The reference has no counterpart: it is a single process
whose only parallelism is the tile loop (reference src/core/Utils/Parallel.h:127-155).
"""
import numpy as np


def shard_range(spp_total, rank, world_size):
    """Contiguous, balanced sample-index range [begin, end) of `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    # the library's own rule (rt_shard_range; tests/test_abi.py pins the two as equal), in plain integers so that a rank
    # without the device library (a CPU-only host in a gloo group) can still find its shard
    return rank * spp_total // world_size, (rank + 1) * spp_total // world_size


def reduce_film(rgb, weight, dst=0):
    """Sum-reduces torch tensors rgb (H,W,3) and weight (H,W) to rank `dst` (in place on dst). No-op without an
    initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rgb, weight
    # one exchange step: rgb sums and weight sums travel as one packed buffer
    import torch
    packed = torch.cat([rgb.reshape(-1), weight.reshape(-1)])
    dist.reduce(packed, dst=dst, op=dist.ReduceOp.SUM)
    if dist.get_rank() == dst:
        rgb.copy_(packed[:rgb.numel()].view_as(rgb))
        weight.copy_(packed[rgb.numel():].view_as(weight))
    return rgb, weight


def render_sharded(render_range, spp_total, rank=None, world_size=None, device="cpu", dst=0):
    """Runs `render_range(spp_begin, spp_end) -> (rgb ndarray or tensor, weight)` for this rank's shard and
    reduces to rank `dst`. Returns torch tensors (meaningful on rank dst)."""
    import torch
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    b, e = shard_range(spp_total, rank, world_size)
    rgb, w = render_range(b, e)
    if isinstance(rgb, np.ndarray):
        rgb = torch.from_numpy(np.ascontiguousarray(rgb)).to(device)
        w = torch.from_numpy(np.ascontiguousarray(w)).to(device)
    return reduce_film(rgb, w, dst)
