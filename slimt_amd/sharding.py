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


def cpu_set_for_rank(local_rank: int, local_world: int, available=None) -> List[int]:
    """The host cores of one rank: the process's allowed cores (`available`, default os.sched_getaffinity(0)) cut into
    `local_world` contiguous, disjoint, equally sized runs; rank r takes run r. A rank runs ~20 worker threads that
    queue launches and wait on streams (slimt's Async workers, Frontend.cc:207-227): eight ranks whose threads
    wander over one host's cores disturb each other's launch latency, which is all the host contributes. With fewer
    cores than ranks every rank gets one core (shared by neighbours)."""
    import os
    cores = sorted(available if available is not None else os.sched_getaffinity(0))
    if local_world <= 0 or not 0 <= local_rank < local_world or not cores:
        raise ValueError("bad rank / world / core set")
    per = len(cores) // local_world
    if per == 0:
        return [cores[local_rank % len(cores)]]
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank: int, local_world: int) -> List[int]:
    """Apply cpu_set_for_rank to this process (before anything touches the GPU or starts a thread: new threads inherit
    it). Returns the set; where the platform has no affinity call, the unchanged set."""
    import os
    mine = cpu_set_for_rank(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else mine
    return mine


def device_identity(index: int) -> dict:
    """PCI bus id and NUMA node of HIP device `index` as the runtime and sysfs report them (None where they do not)."""
    info = {"device": index, "pci_bus_id": None, "numa_node": None}
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        if hasattr(p, "pci_bus_id"):
            info["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
            try:
                with open(f"/sys/bus/pci/devices/{info['pci_bus_id']}/numa_node") as f:
                    info["numa_node"] = int(f.read().strip())
            except (OSError, ValueError):
                pass
    except Exception:  # a report only: never a reason to fail the run
        pass
    return info


def count_ranks(dist, device) -> int:
    """SUM of ones over the ranks: how many processes really took part (1 without a process group)."""
    if dist is None:
        return 1
    import torch
    k = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.SUM)
    return int(k.item())
