"""LMCRemoteBackend -- serialise -> connector.set / connector.get -> deserialise.

Mirror of lmcache/storage_backend/remote_backend.py:23-180: the ONLY caller of
the CacheGen serde in the reference (put_blocking :119-126, get :154-168), with
the same async-put worker thread (:51-69) and existing-keys cache (:111-117).
The pipelined variant (:183-275) is a "next" row.
"""
import queue
import threading
from typing import List, Optional, Set

import torch

from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.abstract_backend import LMCBackendInterface
from lmcache_amd.storage_backend.connector import CreateConnector
from lmcache_amd.storage_backend.serde import CreateSerde
from lmcache_amd.utils import CacheEngineKey, _lmcache_nvtx_annotate

logger = init_logger(__name__)


class RemoteBackendEndSignal:
    pass


class LMCRemoteBackend(LMCBackendInterface):
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        super().__init__()
        self.existing_keys: Set[CacheEngineKey] = set()
        self.put_thread = None
        assert config.remote_url is not None, "Need to provide remote_url when using LMCRemoteBackend"
        assert config.remote_serde is not None, "Need to provide remote_serde when using LMCRemoteBackend"
        self.connection = CreateConnector(config.remote_url)
        self.serializer, self.deserializer = CreateSerde(config.remote_serde, config, metadata)
        self.dst_device = "cuda"
        self._cuda_device = torch.cuda.current_device() if torch.cuda.is_available() else None
        self.put_queue: "queue.Queue" = queue.Queue()
        self.put_thread = threading.Thread(target=self.put_worker, daemon=True)
        self.put_thread.start()

    @_lmcache_nvtx_annotate
    def put_worker(self):
        if self._cuda_device is not None:
            torch.cuda.set_device(self._cuda_device)
        while True:
            item = self.put_queue.get()
            if isinstance(item, RemoteBackendEndSignal):
                break
            key, value = item
            try:
                self.put_blocking(key, value)
            except Exception:
                logger.exception("asynchronous remote put failed")

    def list(self) -> List[CacheEngineKey]:
        keys = [CacheEngineKey.from_string(k) for k in self.connection.list()]
        self.existing_keys.update(keys)
        return keys

    def contains(self, key: CacheEngineKey) -> bool:
        if key in self.existing_keys:
            return True
        if self.connection.exists(key.to_string()):
            self.existing_keys.add(key)
            return True
        return False

    def put_blocking(self, key: CacheEngineKey, kv_chunk: torch.Tensor) -> None:
        self.connection.set(key.to_string(), self.serializer.to_bytes(kv_chunk))
        self.existing_keys.add(key)

    def put(self, key: CacheEngineKey, kv_chunk: torch.Tensor, blocking: bool = True) -> None:
        if blocking:
            self.put_blocking(key, kv_chunk)
        else:
            self.put_queue.put((key, kv_chunk))

    @_lmcache_nvtx_annotate
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        if not self.contains(key):
            return None
        bs = self.connection.get(key.to_string())
        if bs is None or len(bs) == 0:
            return None
        return self.deserializer.from_bytes(bs).to(self.dst_device)

    def close(self):
        if self.put_thread is not None and self.put_thread.is_alive():
            self.put_queue.put(RemoteBackendEndSignal())
            self.put_thread.join()
        if self.connection is not None:
            self.connection.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
