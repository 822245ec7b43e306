"""Sharding samples-per-pixel over the GPUs of a node.

The scene is replicated; rank g of G renders frame indices g, g+G, g+2G, ...
into an un-normalised sum (ACCUM_SUM), so that the union over ranks is exactly
the single-GPU frame schedule. One RCCL reduce (torch.distributed, backend
nccl) sums the RGBA32F accumulators onto rank 0, which tone-maps once. No
collective is needed while rendering.
"""
from __future__ import annotations


def frame_schedule(rank: int, world: int, first_frame: int, spp_per_rank: int):
    """(first frame index, stride, count) of the dispatches rank renders."""
    if not (0 <= rank < world) or spp_per_rank < 0:
        raise ValueError("bad rank / world / spp")
    return first_frame + rank, world, spp_per_rank


def frames_of(rank: int, world: int, first_frame: int, spp_per_rank: int):
    f0, stride, n = frame_schedule(rank, world, first_frame, spp_per_rank)
    return [f0 + k * stride for k in range(n)]


def split_spp(total_spp: int, world: int):
    """Samples per rank when total_spp does not divide evenly: the first
    total_spp % world ranks take one more."""
    base, rem = divmod(total_spp, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def reduce_sum(accum, dst: int = 0):
    """Sum the per-rank accumulators (torch tensor [H, W, 4] float32) onto dst."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def normalise(accum_sum):
    """Sum image -> mean image (rgb / alpha, alpha = sample count)."""
    out = accum_sum.clone()
    out[..., :3] = accum_sum[..., :3] / accum_sum[..., 3:4]
    out[..., 3] = 1.0
    return out
