"""CacheEngineKey, KVCache and the range-annotation decorator.

Mirror of the reference's lmcache/utils.py for the hot path's host side:
  * CacheEngineKey + to_string/from_string     utils.py:12-39
  * _lmcache_nvtx_annotate                      utils.py:42-60 (NVTX there; here a
    roctx range when torch exposes one, otherwise the identity -- so profiles
    taken with rocprofv3 --marker-trace carry the same range names)
"""
import functools
from dataclasses import dataclass
from typing import Tuple

import torch

# nested tuple handed to LMCacheEngine.store / returned by retrieve:
# per layer (K, V), each [T,H,D] ("vllm") or [H,T,D] ("huggingface")
KVCache = Tuple[Tuple[torch.Tensor, torch.Tensor], ...]

_KEY_FIELDS = 5


@dataclass(frozen=True)
class CacheEngineKey:
    """Identity of one stored chunk.  world_size and worker_id are part of the key so that
    every tensor-parallel rank caches its own KV-head shard independently (utils.py:12-31)."""
    fmt: str
    model_name: str
    world_size: int
    worker_id: int
    chunk_hash: str

    def to_string(self) -> str:
        return "@".join((self.fmt, self.model_name, str(self.world_size), str(self.worker_id), self.chunk_hash))

    @staticmethod
    def from_string(s: str) -> "CacheEngineKey":
        parts = s.split("@")
        if len(parts) != _KEY_FIELDS:
            raise ValueError(f"Invalid key string: {s}")
        fmt, model, ws, wid, h = parts
        return CacheEngineKey(fmt, model, int(ws), int(wid), h)


def _range_push_pop():
    nv = getattr(getattr(torch, "cuda", None), "nvtx", None)  # roctx-backed on ROCm builds
    if nv is None or not torch.cuda.is_available():
        return None
    return nv.range_push, nv.range_pop


def _lmcache_nvtx_annotate(func, domain: str = "lmcache"):
    """Wrap `func` in a named profiler range (roctx on ROCm).  Identity when no GPU is present."""
    name = f"{domain}:{func.__qualname__}"

    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        pp = _range_push_pop()
        if pp is None:
            return func(*args, **kwargs)
        pp[0](name)
        try:
            return func(*args, **kwargs)
        finally:
            pp[1]()

    return wrapped
