"""Lossless torch.save/torch.load serde (mirror of
lmcache/storage_backend/serde/torch_serde.py:11-31).  BASELINE config 1's
"CPU torch serde": kept as the lossless baseline the engine tests compare with
bit equality; no GPU work of its own."""
import io

import torch

from lmcache_amd.storage_backend.serde.serde import Deserializer, Serializer


class TorchSerializer(Serializer):
    def __init__(self):
        super().__init__()

    def to_bytes(self, t: torch.Tensor) -> bytes:
        buf = io.BytesIO()
        torch.save(t.detach().cpu().clone(), buf)
        return buf.getvalue()


class TorchDeserializer(Deserializer):
    def __init__(self):
        super().__init__()

    def from_bytes_normal(self, b: bytes) -> torch.Tensor:
        return torch.load(io.BytesIO(bytes(b)), weights_only=True)

    def from_bytes(self, b: bytes) -> torch.Tensor:
        return self.from_bytes_normal(b)
