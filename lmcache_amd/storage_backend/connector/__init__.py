"""Connector factory (mirror of lmcache/storage_backend/connector/__init__.py:60-102).

The reference's network connectors (lm:// TCP, redis://, redis-sentinel://) move
opaque bytes and are outside the hot path (SURVEY.md section 2 #8, section 8 "out of scope"):
a deployment keeps using the reference's.  Two connectors exist here: the
in-process `mem://` (LMCRemoteBackend + the serde end to end without a server
process) and `xgmi://` (row f1: chunks shared between the instances of one node,
resident in the GPUs' HBM and moved over xGMI -- xgmi_connector.py).
"""
import re
import threading
from typing import Dict, List, Optional

from lmcache_amd.storage_backend.connector.base_connector import RemoteConnector

_STORES: Dict[str, Dict[str, bytes]] = {}
_STORES_LOCK = threading.Lock()


class InProcessConnector(RemoteConnector):
    """`mem://<name>:<port>` -- a process-wide dict of bytes keyed by URL."""

    def __init__(self, name: str):
        with _STORES_LOCK:
            self._store = _STORES.setdefault(name, {})
        self._lock = threading.Lock()

    def exists(self, key: str) -> bool:
        return key in self._store

    def get(self, key: str) -> Optional[bytes]:
        return self._store.get(key)

    def set(self, key: str, obj: bytes) -> None:
        with self._lock:
            self._store[key] = bytes(obj)

    def list(self) -> List[str]:
        return list(self._store.keys())

    def close(self) -> None:
        pass


def CreateConnector(url: str) -> RemoteConnector:
    m = re.match(r"(.*)://(.*):(\d+)", url)
    if not m:
        raise ValueError(f"Invalid remote url {url}")
    scheme = m.group(1)
    if scheme == "mem":
        return InProcessConnector(f"{m.group(2)}:{m.group(3)}")
    if scheme == "xgmi":  # xgmi://<store name>:<ranks sharing it (0: WORLD_SIZE)>: HBM arenas + HIP IPC, row f1
        from lmcache_amd.storage_backend.connector.xgmi_connector import XgmiConnector
        return XgmiConnector(m.group(2), int(m.group(3)))
    if scheme in ("lm", "redis", "redis-sentinel"):
        raise ValueError(f"{scheme}:// connectors are network I/O outside lmcache_amd's scope "
                         f"(use the reference's lmcache.storage_backend.connector)")
    raise ValueError(f"Unknown connector type {scheme} (url is: {url})")
