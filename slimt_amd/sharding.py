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


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def core_topology(cores, sysfs="/sys/devices/system/cpu"):
    """(package, physical core) of every logical core, from sysfs; a core whose topology files are missing is its own
    physical core on package 0 (containers, other platforms)."""
    topo = {}
    for c in cores:
        try:
            with open(f"{sysfs}/cpu{c}/topology/physical_package_id") as f:
                pkg = int(f.read())
            with open(f"{sysfs}/cpu{c}/topology/core_id") as f:
                core = int(f.read())
            topo[c] = (pkg, core)
        except (OSError, ValueError):
            topo[c] = (0, 1 << 20 | c)
    return topo


def gpu_local_cpulists(sysfs="/sys/bus/pci/drivers/amdgpu"):
    """The host cores next to every amdgpu device (its PCI function's local_cpulist), in PCI address order -- the order
    the HIP runtime enumerates devices in unless ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES reorder them. [] when sysfs
    has none (no GPU, a container without the driver's sysfs)."""
    import os
    out = []
    try:
        names = sorted(n for n in os.listdir(sysfs) if ":" in n)
    except OSError:
        return out
    for n in names:
        try:
            with open(f"{sysfs}/{n}/local_cpulist") as f:
                out.append(_parse_cpulist(f.read()))
        except (OSError, ValueError):
            return []
    return out


def cpu_set_for_rank(local_rank: int, local_world: int, available=None, topology=None, gpu_cpus=None) -> List[int]:
    """The host cores of one rank: a disjoint, (near-)equal share of the process's allowed cores (`available`, default
    os.sched_getaffinity(0)). A rank runs ~20 worker threads that queue launches and wait on streams (slimt's Async
    workers, Frontend.cc:207-227): eight ranks whose threads wander over one host's cores disturb each other's launch
    latency, which is all the host contributes.

    Topology: logical cores are ordered by (package, physical core, id), so SMT siblings stay with one rank
    and a rank's run does not straddle sockets more than it must; and when sysfs lists exactly one amdgpu device per
    local rank (or more: the first `local_world` of them) and they are not reordered by *_VISIBLE_DEVICES (`gpu_cpus`: their local_cpulist in PCI order), rank r
    takes its share of the cores NEXT TO GPU r -- the ranks whose GPUs hang off the same socket split that socket's
    cores. With fewer cores than ranks every rank gets one core (shared by neighbours)."""
    import os
    cores = sorted(available if available is not None else os.sched_getaffinity(0))
    if local_world <= 0 or not 0 <= local_rank < local_world or not cores:
        raise ValueError("bad rank / world / core set")
    topo = topology if topology is not None else core_topology(cores)
    order = sorted(cores, key=lambda c: (topo.get(c, (0, c)), c))
    if gpu_cpus and len(gpu_cpus) >= local_world:
        gpu_cpus = gpu_cpus[:local_world]  # (N ranks on a node of more GPUs use devices 0 .. N - 1)
        # ranks grouped by identical neighbourhoods; each group splits its neighbourhood's allowed cores
        mine = [c for c in order if c in set(gpu_cpus[local_rank])]
        group = [r for r in range(local_world) if sorted(gpu_cpus[r]) == sorted(gpu_cpus[local_rank])]
        if len(mine) >= len(group):
            per = len(mine) // len(group)
            k = group.index(local_rank)
            return sorted(mine[k * per:(k + 1) * per])
    per = len(order) // local_world
    if per == 0:
        return [order[local_rank % len(order)]]
    return sorted(order[local_rank * per:(local_rank + 1) * per])


def pin_rank(local_rank: int, local_world: int) -> List[int]:
    """Apply cpu_set_for_rank to this process (before anything touches the GPU or starts a thread: new threads inherit
    it). Returns the set; where the platform has no affinity call, the unchanged set. The GPUs' neighbourhoods are used
    only when no *_VISIBLE_DEVICES variable may have reordered or hidden devices."""
    import os
    reordered = any(os.environ.get(v) for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    mine = cpu_set_for_rank(local_rank, local_world, gpu_cpus=None if reordered else gpu_local_cpulists())
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
