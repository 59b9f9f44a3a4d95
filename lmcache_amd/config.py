"""Engine configuration and metadata -- field-for-field the reference's
lmcache/config.py so that YAML files, LMCACHE_CONFIG_FILE and constructor
calls written for LMCache keep working:

  LMCacheEngineMetadata(model_name, world_size, worker_id, fmt, dtype)      config.py:8-19
  LMCacheEngineConfig(chunk_size, local_device, remote_url, remote_serde,
                      pipelined_backend, save_decode_cache)                  config.py:22-31
  .from_defaults / .from_legacy / .from_file                                 config.py:34-124
  GlobalConfig.enable_debug                                                  config.py:130-139

One field is added at the END (default None, so positional construction is
unchanged): `local_serde`.  local_device="cpu" + local_serde="cachegen" selects
the MI355X path that keeps CacheGen-encoded chunks in pinned host DRAM
(BASELINE.json north star) instead of raw tensors.
"""
import re
from dataclasses import dataclass
from typing import Optional

import yaml

_DISK_RE = re.compile(r"file://(.*)/")
_URL_RE = re.compile(r"(.*)://(.*):(\d+)")


@dataclass
class LMCacheEngineMetadata:
    model_name: str   # name of the LLM
    world_size: int   # tensor-parallel world size
    worker_id: int    # this rank
    fmt: str          # "vllm" | "huggingface"
    dtype: str        # dtype of the KV tensors (informational, as in the reference)


@dataclass
class LMCacheEngineConfig:
    chunk_size: int
    local_device: Optional[str]
    remote_url: Optional[str]
    remote_serde: Optional[str]  # "torch" | "cachegen"
    pipelined_backend: bool
    save_decode_cache: bool
    local_serde: Optional[str] = None  # None (raw tensors) | "cachegen" (encoded chunks in pinned DRAM)

    @staticmethod
    def from_defaults(chunk_size: int = 256, local_device: str = "cuda",
                      remote_url: str = "redis://localhost:6379", remote_serde: str = "torch",
                      pipelined_backend: bool = False, save_decode_cache: bool = False,
                      local_serde: Optional[str] = None) -> "LMCacheEngineConfig":
        return LMCacheEngineConfig(chunk_size, local_device, remote_url, remote_serde, pipelined_backend,
                                   save_decode_cache, local_serde)

    @staticmethod
    def from_legacy(chunk_size: int = 256, backend: str = "cuda", persist_path: Optional[str] = None,
                    remote_serde: Optional[str] = "torch", pipelined_backend: bool = False,
                    save_decode_cache: bool = False, local_serde: Optional[str] = None) -> "LMCacheEngineConfig":
        """`backend` is "cpu" | "cuda" | "file://<dir>/" | "<scheme>://host:port" (config.py:52-82)."""
        local_device: Optional[str] = None
        remote_url: Optional[str] = None
        if backend in ("cpu", "cuda"):
            local_device = backend
        elif _DISK_RE.match(backend):
            local_device = backend[len("file://"):]
        elif _URL_RE.match(backend):
            remote_url = backend
        return LMCacheEngineConfig(chunk_size, local_device, remote_url, remote_serde, pipelined_backend,
                                   save_decode_cache, local_serde)

    @staticmethod
    def from_file(file_path: str) -> "LMCacheEngineConfig":
        """YAML loader with the reference's keys, defaults and validation (config.py:85-124)."""
        with open(file_path, "r") as fin:
            raw = yaml.safe_load(fin) or {}
        local_device = raw.get("local_device", None)
        remote_url = raw.get("remote_url", None)
        if local_device not in ("cpu", "cuda", None):
            if not (isinstance(local_device, str) and _DISK_RE.match(local_device)):
                raise ValueError(f"Invalid local storage device: {local_device}")
            local_device = local_device[len("file://"):]
        if remote_url is not None and not (isinstance(remote_url, str) and _URL_RE.match(remote_url)):
            raise ValueError(f"Invalid remote storage url: {remote_url}")
        return LMCacheEngineConfig(raw.get("chunk_size", 256), local_device, remote_url,
                                   raw.get("remote_serde", "torch"), raw.get("pipelined_backend", False),
                                   raw.get("save_decode_cache", False), raw.get("local_serde", None))


class GlobalConfig:
    """Process-wide switches (config.py:130-139).  Debug wrappers time serde calls."""
    enable_debug: bool = True

    @classmethod
    def set_debug(cls, enable: bool) -> None:
        cls.enable_debug = enable

    @classmethod
    def is_debug(cls) -> bool:
        return cls.enable_debug
