"""Multi-GPU sharding of the hot path.

Every (chunk, plane, channel-group) stream is independent and chunks are
independently keyed, so the path shards by chunk with NO data-path collective:
rank r of W encodes / decodes / offloads chunks {i : i mod W == r} (replicated
instances), or its own KV-head shard under a key that embeds (world_size,
worker_id) (tensor parallel; lmcache/utils.py:12-31).  torch.distributed is
used for control only: barriers and the max-over-ranks timing of bench.py.
Key ownership for sharing across instances (row f1, "next") is H(key) mod W.
"""
import hashlib
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_chunks(nchunks: int, rank: int, world: int) -> List[int]:
    """Round-robin chunk ownership."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, nchunks, world))


def owner_rank(key_string: str, world: int) -> int:
    """Stable owner of a chunk key (independent of PYTHONHASHSEED)."""
    return int.from_bytes(hashlib.sha256(key_string.encode("utf-8")).digest()[:8], "little") % world


def max_over_ranks(value: float, device: torch.device) -> float:
    """Max of a per-rank scalar (the timed region's elapsed seconds)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_rate(units_per_rank: Sequence[float], elapsed_max: float) -> float:
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    return sum(units_per_rank) / elapsed_max


def gpu_numa_node(pci_bus_id: str, sysfs: str = "/sys/bus/pci/devices") -> Optional[int]:
    """NUMA node of a GPU from sysfs (None when the platform does not say)."""
    try:
        with open(os.path.join(sysfs, pci_bus_id.lower(), "numa_node")) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except (OSError, ValueError):
        return None


def numa_cpus(node: int, sysfs: str = "/sys/devices/system/node") -> List[int]:
    """CPUs of a NUMA node ("0-31,64-95" style cpulist)."""
    try:
        with open(os.path.join(sysfs, f"node{node}", "cpulist")) as f:
            text = f.read().strip()
    except OSError:
        return []
    cpus: List[int] = []
    for part in text.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(local_rank: int) -> Optional[int]:
    """Run this process (and the threads it starts) on the NUMA node its GPU hangs off, so that the pinned host
    arenas it first-touches -- the offload target of this rank -- are local to that GPU's PCIe root: with eight
    ranks offloading at once, no blob crosses the inter-socket link.  Best effort: returns the node, or None when
    the platform exposes no topology (then nothing changes)."""
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bus = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        node = gpu_numa_node(bus)
        if node is None:
            return None
        allowed = set(os.sched_getaffinity(0))
        cpus = [c for c in numa_cpus(node) if c in allowed]
        if cpus:
            os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None
