"""Multi-GPU plumbing of the benchmark / chunk driver: genome chunks are independent
(phase.c:276-473), so ranks shard chunks and never exchange data on the hot path.  The only
communication is the benchmark's barrier and the max/sum reduction of (elapsed, units)."""
from __future__ import annotations

from typing import List, Tuple


def chunk_seeds(rank: int, n_chunks: int) -> List[int]:
    """Disjoint seeds per rank (weak scaling: every rank owns n_chunks chunks)."""
    return [1000 * rank + i + 1 for i in range(n_chunks)]


def shard_chunks(n_total: int, rank: int, world: int) -> List[int]:
    """Strong-scaling split of a fixed chunk list: chunk i goes to rank i % world, the order the
    reference's dynamic OpenMP schedule would hand them out for equal-cost chunks."""
    return [i for i in range(n_total) if i % world == rank]


def reduce_elapsed_and_units(dist, elapsed: float, units: float, device=None) -> Tuple[float, float]:
    """MAX over ranks of the timed region, SUM over ranks of the units processed."""
    if dist is None:
        return elapsed, units
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
