"""Data-parallel gradient averaging: one process per GPU, RCCL all-reduce over xGMI (backend "nccl" on ROCm).

The reference is single-GPU (scripts/train/train_nersemble.py:272-274 hard-codes world_size = 1); rays are
independent until the loss mean, so each rank draws its own rays and holds a full replica; the only exchange is
the sum of gradients.  Large buffers (the 1.6 GB hash-table gradient) go out as their own
collective; small tensors are flattened into one bucket so a step issues O(1) collectives.
"""
from typing import Iterable

import torch
import torch.distributed as dist

SMALL_BUCKET_ELEMS = 1 << 22


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int, group=None) -> None:
    """In-place average of ``p.grad`` over all ranks.  Parameters without a gradient contribute zeros (every rank
    must join every collective)."""
    if world_size <= 1:
        return
    params = [p for p in params if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    big = [p for p in params if p.grad.numel() >= SMALL_BUCKET_ELEMS]
    small = [p for p in params if p.grad.numel() < SMALL_BUCKET_ELEMS]
    handles = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True) for p in big]
    flat = None
    if small:
        flat = torch.cat([p.grad.reshape(-1).float() for p in small])
        handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for h in handles:
        h.wait()
    inv = 1.0 / world_size
    for p in big:
        p.grad.mul_(inv)
    if small:
        flat.mul_(inv)
        off = 0
        for p in small:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def shard_seed(base_seed: int, rank: int) -> int:
    """Rank-specific RNG stream for ray sampling (identical model init comes from the un-sharded base seed)."""
    return base_seed + 7919 * rank
