"""Pin a rank's host threads to its GPU's NUMA-local cores.

Eight ranks of one node each issue ~170 eager launches per 5 ms step from one Python thread (the reference starts them with
mp.spawn, tools/train_stand.py:151-155, and pins nothing).  Unpinned, seven sibling launch loops stretch an eager step by ~20 %
(tools/host_contention_probe.py): the kernel's scheduler migrates the launch thread between cores and sockets, away from the
GPU's PCIe root.  `pin_rank` gives every local rank its own slice of the cores of the NUMA node its GPU hangs on
(/sys/bus/pci/devices/<bdf>/local_cpulist), or -- where the platform states no locality -- an even slice of the process's
allowed cores.  Host-side only; inherited by the DataLoader workers and RCCL's proxy threads the rank starts afterwards.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_local_cores(device_index: int) -> Optional[List[int]]:
    """cores of the NUMA node the HIP device hangs on, or None when sysfs does not say (numa_node = -1, containers without /sys/bus/pci)"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(base, "local_cpulist")) as f:
            cores = _parse_cpulist(f.read())
        return cores or None
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def plan(local_rank: int, local_world: int, allowed: List[int], local_cores_by_rank: Dict[int, Optional[List[int]]]) -> List[int]:
    """the cores of `local_rank`: ranks whose GPUs share a NUMA node split that node's (allowed) cores evenly, in rank order; a rank
    without locality information takes an even slice of all allowed cores.  Never empty; never outside `allowed`."""
    allowed = sorted(allowed)
    mine = local_cores_by_rank.get(local_rank)
    if mine:
        pool = [c for c in sorted(mine) if c in set(allowed)]
        peers = [r for r in range(local_world) if local_cores_by_rank.get(r) and sorted(local_cores_by_rank[r]) == sorted(mine)]
    else:
        pool, peers = [], []
    if not pool:
        pool, peers = allowed, list(range(local_world))
    k = max(1, len(pool) // max(len(peers), 1))
    i = peers.index(local_rank) if local_rank in peers else local_rank % max(len(pool) // k, 1)
    cores = pool[i * k:(i + 1) * k]
    return cores or pool


def pin_rank(local_rank: int, local_world: int, device_index: Optional[int] = None, enable: bool = True) -> dict:
    """sched_setaffinity of THIS process; returns what was done (bench.py prints it).  enable = False: report only."""
    if not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "reason": "no sched_setaffinity on this platform"}
    allowed = sorted(os.sched_getaffinity(0))
    if local_world <= 1 or not enable:
        return {"pinned": False, "reason": "single rank" if local_world <= 1 else "disabled", "allowed_cores": len(allowed)}
    if len(allowed) < 4 * local_world:
        # a rank runs its launch thread, RCCL's proxy thread and its DataLoader workers: with fewer than four cores each, slices would
        # serialise them -- leave the scheduler alone
        return {"pinned": False, "reason": f"{len(allowed)} allowed cores for {local_world} ranks", "allowed_cores": len(allowed)}
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    by_rank = {r: (gpu_local_cores(r % ndev) if ndev else None) for r in range(local_world)}
    if device_index is not None and ndev:
        by_rank[local_rank] = gpu_local_cores(device_index)
    cores = plan(local_rank, local_world, allowed, by_rank)
    os.sched_setaffinity(0, cores)
    return {"pinned": True, "cores": len(cores), "first_core": cores[0], "last_core": cores[-1], "numa_local": bool(by_rank.get(local_rank)),
            "allowed_cores": len(allowed)}
