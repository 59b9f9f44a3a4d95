"""LMCHybridBackend -- a local tier in front of a remote one (mirror of the reference's
lmcache/storage_backend/hybrid_backend.py:17-116): writes go to both, reads try the local tier first and fill
it from the remote one, and at start-up the chunks the remote store already holds for this model / rank are
pulled into the local tier (:41-66).

Orchestration only -- every byte still moves through the two backends it wraps (LMCLocalBackend: HBM or pinned
host DRAM; LMCRemoteBackend / LMCPipelinedRemoteBackend: serde + connector, e.g. xgmi://), so the HIP hot path
is the same.  It speaks the chunk-tensor protocol of LMCBackendInterface; the engine gathers / scatters chunks
with one lmc_copy_kv pass each, as for any backend without the range protocol.
"""
import time
from typing import Iterable, List, Optional

import torch

from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.abstract_backend import LMCBackendInterface
from lmcache_amd.storage_backend.local_backend import LMCLocalBackend
from lmcache_amd.storage_backend.remote_backend import LMCPipelinedRemoteBackend, LMCRemoteBackend
from lmcache_amd.utils import CacheEngineKey, _lmcache_nvtx_annotate

logger = init_logger(__name__)


class LMCHybridBackend(LMCBackendInterface):
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        super().__init__()
        self.local_store = LMCLocalBackend(config, metadata)
        remote_cls = LMCPipelinedRemoteBackend if config.pipelined_backend else LMCRemoteBackend
        self.remote_store = remote_cls(config, metadata)
        self.remote_store.supports_kv_layout = False  # chunk tensors between the two tiers
        self._warm_up(metadata)

    def _warm_up(self, metadata: LMCacheEngineMetadata) -> None:
        """Pull what the remote store already has for this (model, world size, rank) into the local tier."""
        t0 = time.perf_counter()
        pulled = 0
        keys = self.remote_store.list()
        for key in keys:
            mine = (metadata.model_name, metadata.world_size, metadata.worker_id)
            if (key.model_name, key.world_size, key.worker_id) != mine:
                continue
            chunk = self.remote_store.get(key)
            if chunk is not None:
                self.local_store.put(key, chunk)
                pulled += 1
        logger.info("hybrid backend: %d of %d remote chunks pulled into the local tier in %.2f s", pulled, len(keys),
                    time.perf_counter() - t0)

    def contains(self, key: CacheEngineKey) -> bool:
        return self.local_store.contains(key) or self.remote_store.contains(key)

    def put(self, key: CacheEngineKey, value: torch.Tensor, blocking: bool = True) -> None:
        # write-through: the local copy is there when put returns, the remote one follows `blocking`
        self.local_store.put(key, value, blocking=True)
        self.remote_store.put(key, value, blocking)

    @_lmcache_nvtx_annotate
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        chunk = self.local_store.get(key)
        if chunk is None:
            chunk = self.remote_store.get(key)
            if chunk is not None:
                self.local_store.put(key, chunk)  # read-through fill
        return chunk

    @_lmcache_nvtx_annotate
    def batched_get(self, keys: Iterable[CacheEngineKey]) -> List[Optional[torch.Tensor]]:
        keys = list(keys)
        out: List[Optional[torch.Tensor]] = [self.local_store.get(k) for k in keys]
        missing = [i for i, c in enumerate(out) if c is None]
        if missing:
            fetched = list(self.remote_store.batched_get(keys[i] for i in missing))
            for i, chunk in zip(missing, fetched):
                if chunk is not None:
                    self.local_store.put(keys[i], chunk)
                    out[i] = chunk
        return out

    def close(self):
        self.local_store.close()
        self.remote_store.close()
