"""Data-parallel plumbing shared by the trainer and bench: one process per GPU, batch sharded by rank,
ONE flat gradient bucket all-reduced per step (RCCL when the tensors are on GPUs: backend "nccl" is RCCL
on ROCm; gloo on CPU in the tests).  The reference has no distributed code; the
semantics follow torch DDP defaults: gradients are averaged over ranks, BatchNorm uses per-replica batch
statistics, and BatchNorm buffers are broadcast from rank 0 before each forward.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


LOCAL = "local"   # process_group value of a trainer that must not communicate although a default group exists


def world(group=None):
    if group is LOCAL:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(total, rank, world_size):
    """Contiguous shard [lo, hi) of `total` items for `rank`; the first `total % world_size` ranks get one extra."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_flat_sum(flat: torch.Tensor, group=None, async_op=False):
    """SUM all-reduce of the single flat gradient bucket.  Returns (work or None, grad_scale): the caller
    folds grad_scale = 1/world into the optimizer kernel instead of spending a pass on the division."""
    rank, ws = world(group)
    if ws == 1:
        return None, 1.0
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work, 1.0 / ws


def broadcast_buffers(buf: torch.Tensor, group=None, src=0):
    """DDP default broadcast_buffers=True: rank 0's BatchNorm running statistics win."""
    _, ws = world(group)
    if ws > 1:
        dist.broadcast(buf, src=src, group=group)


def max_over_ranks(value: float, device, group=None) -> float:
    _, ws = world(group)
    if ws == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
