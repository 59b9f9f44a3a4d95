"""CacheGenDeserializer -- drop-in for the reference's
lmcache/storage_backend/serde/cachegen_decoder.py:108-202 (same constructor,
same `from_bytes(bytes) -> Tensor`): decode_function_gpu / decode_chunk /
torchac_cuda.decode_fast_prefsum, the uint8->fp32 inflation, do_dequantize and
the stack/reshape/permute/cast tail run as ONE fused HIP kernel that writes the
16-bit result in its final layout.

Output contract kept from the reference (:190-200): "vllm" ->
[L,2,T,H,D] bfloat16, "huggingface" -> [L,2,H,T,D] float16, on the current
CUDA device (the reference returns a permuted view; we return the same values
contiguous).
"""
import threading

import torch

from lmcache_amd import native
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
from lmcache_amd.storage_backend.serde.cachegen_device import get_codec
from lmcache_amd.storage_backend.serde.serde import Deserializer
from lmcache_amd.utils import _lmcache_nvtx_annotate

logger = init_logger(__name__)


def output_spec(fmt: str, L: int, T: int, H: int, D: int):
    """Shape and dtype rule of cachegen_decoder.py:190-200."""
    if fmt == "vllm":
        return (L, 2, T, H, D), torch.bfloat16
    if fmt == "huggingface":
        return (L, 2, H, T, D), torch.float16
    raise RuntimeError("Unknown format %s" % fmt)


class CacheGenDeserializer(Deserializer):
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        native.lib()
        self.cachegen_config = CacheGenConfig.from_model_name(metadata.model_name)
        self.chunk_size = config.chunk_size
        self.fmt = metadata.fmt
        self.key_bins = self.cachegen_config.key_bins()
        self.value_bins = self.cachegen_config.value_bins()
        self._lock = threading.Lock()

    def make_key_bins(self, config: CacheGenConfig) -> torch.Tensor:
        """Bins per key layer as the reference holds them: a float32 tensor (cachegen_decoder.py:121-126; on the host
        here -- the decoder reads the bins from the blob, this mirror only serves callers that ask for the tensor)."""
        return torch.tensor(config.key_bins(), dtype=torch.float32)

    def make_value_bins(self, config: CacheGenConfig) -> torch.Tensor:
        """... and per value layer (cachegen_decoder.py:128-132)."""
        return torch.tensor(config.value_bins(), dtype=torch.float32)

    @_lmcache_nvtx_annotate
    def from_bytes(self, bs) -> torch.Tensor:
        h = native.blob_info(bs)  # validates magic / geometry / length on the host
        shape, dtype = output_spec(self.fmt, h.num_layers, h.ntokens, h.num_heads, h.head_size)
        with self._lock:
            dev = torch.cuda.current_device()
            out = torch.empty(shape, dtype=dtype, device=torch.device("cuda", dev))
            codec = get_codec(dev)
            job = codec.decode([bs], native.KVLayout.from_chunk(out, self.fmt), 0, int(h.ntokens))
            # the caller owns `out` and may use it on any stream: finish (this decode only) before returning;
            # a blob whose streams do not check out raises instead of returning garbage
            codec.finish_decode(job)
            return out
