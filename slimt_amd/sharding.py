"""Data-parallel sharding of independent sentences over ranks (one process per
GPU, replicated weights, NO collective on the data path -- sentences never
interact, Frontend.cc:212-226 / SURVEY 8e). torch.distributed is used only to
line ranks up and to reduce the measured time / token counts."""
from __future__ import annotations

from typing import List, Tuple


def plan_shards(n_sentences: int, batch: int, world: int) -> List[List[Tuple[int, int]]]:
    """Split sentences [0, n) into batches of <= `batch` and deal the batches
    round-robin to `world` ranks. Returns, per rank, a list of (start, count).
    Every sentence is assigned exactly once; ranks differ by at most one batch."""
    if n_sentences < 0 or batch <= 0 or world <= 0:
        raise ValueError("bad sharding arguments")
    plan: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    start, i = 0, 0
    while start < n_sentences:
        count = min(batch, n_sentences - start)
        plan[i % world].append((start, count))
        start += count
        i += 1
    return plan


def reduce_timing(dist, device, seconds: float, tokens: int) -> Tuple[float, int]:
    """MAX of the elapsed time and SUM of the token counts over all ranks
    (dist=None: single process)."""
    if dist is None:
        return seconds, tokens
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    k = torch.tensor([tokens], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.SUM)
    return float(t.item()), int(k.item())
