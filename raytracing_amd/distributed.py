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
    # the library's own rule (rt_shard_range), so that one process per GPU and one context over several GPUs
    # (rt_create_multi) cut a job at the same sample indices
    import ctypes as C
    from .device import device_lib
    b, e = C.c_int(), C.c_int()
    device_lib().rt_shard_range(0, int(spp_total), int(rank), int(world_size), C.byref(b), C.byref(e))
    return b.value, e.value


def reduce_film(rgb, weight, dst=0):
    """Sum-reduces torch tensors rgb (H,W,3) and weight (H,W) to rank `dst` (in place on dst). No-op without an
    initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rgb, weight
    dist.reduce(rgb, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(weight, dst=dst, op=dist.ReduceOp.SUM)
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
