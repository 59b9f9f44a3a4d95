"""Serde factory (mirror of lmcache/storage_backend/serde/__init__.py:19-41)."""
from typing import Tuple

from lmcache_amd.config import GlobalConfig, LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.storage_backend.serde.serde import (Deserializer, DeserializerDebugWrapper, Serializer,
                                                     SerializerDebugWrapper)
from lmcache_amd.storage_backend.serde.torch_serde import TorchDeserializer, TorchSerializer


def CreateSerde(serde_type: str, config: LMCacheEngineConfig,
                metadata: LMCacheEngineMetadata) -> Tuple[Serializer, Deserializer]:
    if serde_type == "torch":
        s, d = TorchSerializer(), TorchDeserializer()
    elif serde_type == "cachegen":
        from lmcache_amd.storage_backend.serde.cachegen_decoder import CacheGenDeserializer
        from lmcache_amd.storage_backend.serde.cachegen_encoder import CacheGenSerializer
        s, d = CacheGenSerializer(config, metadata), CacheGenDeserializer(config, metadata)
    elif serde_type in ("safetensor", "fast"):
        # lossless byte-shuffling serdes of the reference: no GPU work, outside the hot path
        raise ValueError(f"serde type {serde_type!r} is outside lmcache_amd's scope; use the reference's")
    else:
        raise ValueError(f"Invalid serde type: {serde_type}")
    if GlobalConfig.is_debug():
        return SerializerDebugWrapper(s), DeserializerDebugWrapper(d)
    return s, d


__all__ = ["Serializer", "Deserializer", "TorchSerializer", "TorchDeserializer", "CreateSerde"]
