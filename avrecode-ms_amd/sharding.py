"""Slice sharding across ranks (one process per GPU).

Slices are independent units (one coder object each in the reference, recode.cpp:1270, 1525),
so the path shards with no data-path collective: rank r owns a contiguous range of slice
indices and generates / receives only those.  The only collectives are the ones the
measurement needs (barrier, max of the elapsed time, sum of the units).
"""
from __future__ import annotations


def shard_first_slice(rank: int, slices_per_rank: int) -> int:
    """Weak scaling: every rank processes `slices_per_rank` slices; rank r owns
    [r*slices_per_rank, (r+1)*slices_per_rank) of the global slice index space."""
    return rank * slices_per_rank


def shard_range(total_slices: int, rank: int, world: int):
    """Strong-scaling split of `total_slices` into near-equal contiguous ranges."""
    base, extra = divmod(total_slices, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_timing(dist, seconds: float, units: int, device):
    """(max over ranks of `seconds`, sum over ranks of `units`); identity without a process group."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return float(seconds), int(units)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())
