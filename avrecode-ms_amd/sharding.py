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


def lpt_assign(n_bins, world: int):
    """Greedy LPT: slices longest first, each to the rank with the least bins so far (ties: the lower rank).
    Returns the rank of every slice.  Same plan as avr_multi_run (csrc/avr_api.cpp)."""
    order = sorted(range(len(n_bins)), key=lambda i: -int(n_bins[i]))        # stable: equal lengths keep their order
    load = [0] * world
    owner = [0] * len(n_bins)
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        owner[i] = r
        load[r] += int(n_bins[i])
    return owner


def balanced_ranges(n_bins, world: int):
    """Strong-scaling split into CONTIGUOUS slice ranges of near-equal bin totals (what a rank can generate from
    (seed, first_slice) alone): boundary r is where the running total comes closest to r/world of all bins.
    Returns world+1 non-decreasing boundaries, first 0, last len(n_bins)."""
    import numpy as np
    cum = np.cumsum(np.asarray(n_bins, dtype=np.int64))
    total = int(cum[-1]) if len(cum) else 0
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))            # cum[i] is the first total >= target
        if i < len(cum) and (i == 0 or abs(int(cum[i]) - target) <= abs(int(cum[i - 1]) - target)):
            i += 1                                                     # taking slice i too lands closer
        bounds.append(max(bounds[-1], min(i, len(cum))))
    bounds.append(len(cum))
    return bounds


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
