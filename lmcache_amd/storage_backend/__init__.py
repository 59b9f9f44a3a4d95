"""Backend factory (mirror of lmcache/storage_backend/__init__.py:13-44): the same
(local_device, remote_url) -> backend decision table and the same ValueError.

  remote only                -> LMCRemoteBackend (serde + connector), LMCPipelinedRemoteBackend if pipelined_backend
  local "cpu" / "cuda"       -> LMCLocalBackend  (HBM, pinned raw, or pinned CacheGen via local_serde)
  local + remote             -> LMCHybridBackend (write-through / read-through over the two above)
  local path (disk)          -> LMCLocalDiskBackend (raw safetensors files as the reference writes them, or the flat
                                CacheGen blob via local_serde)
"""
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.abstract_backend import LMCBackendInterface

logger = init_logger(__name__)


def CreateStorageBackend(config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata) -> LMCBackendInterface:
    local, remote = config.local_device, config.remote_url
    if local is None and remote is not None:
        from lmcache_amd.storage_backend.remote_backend import LMCPipelinedRemoteBackend, LMCRemoteBackend
        if config.pipelined_backend:  # lmcache/storage_backend/hybrid_backend.py:30-33 makes the same choice
            return LMCPipelinedRemoteBackend(config, metadata)
        return LMCRemoteBackend(config, metadata)
    if local is not None and remote is None:
        if local in ("cpu", "cuda"):
            from lmcache_amd.storage_backend.local_backend import LMCLocalBackend
            return LMCLocalBackend(config, metadata)
        from lmcache_amd.storage_backend.local_backend import LMCLocalDiskBackend
        logger.info(f"Initializing local-only (disk) backend at {local}")
        return LMCLocalDiskBackend(config, metadata)
    if local is not None and remote is not None:
        if local not in ("cpu", "cuda"):  # the reference's hybrid backend only ever builds an LMCLocalBackend (hybrid_backend.py:29)
            raise ValueError(f"hybrid backend: the local tier is 'cpu' or 'cuda', not a path ({local!r})")
        from lmcache_amd.storage_backend.hybrid_backend import LMCHybridBackend
        return LMCHybridBackend(config, metadata)
    raise ValueError(f"Invalid configuration: {config}")
