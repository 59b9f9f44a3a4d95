"""Serializer / Deserializer interfaces (mirror of the reference's
lmcache/storage_backend/serde/serde.py:12-27, 44-57) and the timing wrappers
CreateSerde installs when GlobalConfig.is_debug() (serde.py:30-41, 60-72)."""
import abc
import time
from typing import Union

import torch

from lmcache_amd.logging import init_logger

logger = init_logger(__name__)

BytesLike = Union[bytes, bytearray, memoryview]


class Serializer(abc.ABC):
    @abc.abstractmethod
    def to_bytes(self, t: torch.Tensor) -> bytes:
        """Tensor (any device) -> self-describing bytes."""
        raise NotImplementedError


class Deserializer(abc.ABC):
    @abc.abstractmethod
    def from_bytes(self, bs: BytesLike) -> torch.Tensor:
        """bytes / bytearray (never mutated) -> tensor."""
        raise NotImplementedError


class SerializerDebugWrapper(Serializer):
    def __init__(self, s: Serializer):
        self.s = s

    def to_bytes(self, t: torch.Tensor) -> bytes:
        t0 = time.perf_counter()
        out = self.s.to_bytes(t)
        logger.debug("Serialization took %.2f ms", (time.perf_counter() - t0) * 1e3)
        return out


class DeserializerDebugWrapper(Deserializer):
    def __init__(self, d: Deserializer):
        self.d = d

    def from_bytes(self, bs: BytesLike) -> torch.Tensor:
        t0 = time.perf_counter()
        out = self.d.from_bytes(bs)
        logger.debug("Deserialization took %.2f ms", (time.perf_counter() - t0) * 1e3)
        return out
