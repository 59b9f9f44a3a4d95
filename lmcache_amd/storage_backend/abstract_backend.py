"""LMCBackendInterface -- the storage-backend contract of the reference
(lmcache/storage_backend/abstract_backend.py:12-121): put / contains / get,
default batched_put / batched_get loops, close.

Two OPTIONAL methods extend it for backends that can consume KV where it lies
(no blob materialisation) -- the engine uses them when present and falls back
to the reference's chunk-tensor protocol otherwise:

  put_kv_range(keys, src_layout, fmt, tok_begin, tok_end, chunk_tokens, blocking) -> int   chunks stored
  get_kv_range(keys, dst_layout, fmt, dst_tok0, chunk_tokens) -> int   leading chunks written (a key that has
      gone since `contains` ends the run: "None on a miss", never an exception); raises NativeError when a
      stored blob does not decode -- the engine then reports a miss instead of handing garbage to the model
"""
import abc
from typing import Iterable, Optional, Tuple

import torch

from lmcache_amd.logging import init_logger
from lmcache_amd.utils import CacheEngineKey

logger = init_logger(__name__)


class LMCBackendInterface(abc.ABC):
    # set by backends that implement put_kv_range / get_kv_range
    supports_kv_layout: bool = False

    @abc.abstractmethod
    def put(self, key: CacheEngineKey, kv_chunk: torch.Tensor, blocking: bool = True) -> None:
        """Store one chunk tensor ([L,2,T,H,D] or [L,2,H,T,D]; no batch dimension)."""
        raise NotImplementedError

    @abc.abstractmethod
    def contains(self, key: CacheEngineKey) -> bool:
        raise NotImplementedError

    @abc.abstractmethod
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        """The chunk tensor on the GPU, or None on a miss (never raises on a miss)."""
        raise NotImplementedError

    def batched_put(self, keys_and_chunks: Iterable[Tuple[CacheEngineKey, torch.Tensor]], blocking: bool = True) -> int:
        n = 0
        for key, chunk in keys_and_chunks:
            self.put(key, chunk, blocking=blocking)
            n += 1
        return n

    def batched_get(self, keys: Iterable[CacheEngineKey]) -> Iterable[Optional[torch.Tensor]]:
        for key in keys:
            yield self.get(key) if self.contains(key) else None

    @abc.abstractmethod
    def close(self):
        pass
