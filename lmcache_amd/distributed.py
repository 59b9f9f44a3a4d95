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
from typing import List, Sequence

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
