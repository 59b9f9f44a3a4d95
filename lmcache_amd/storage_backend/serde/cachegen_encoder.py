"""CacheGenSerializer -- drop-in for the reference's
lmcache/storage_backend/serde/cachegen_encoder.py:328-389 (same constructor,
same `to_bytes(Tensor) -> bytes`), running the whole of encode_function
(:266-325: _split_kv, torch_quant_vectorized x2, torchac_cuda.calculate_cdf x2,
encode_fast_new, collect_bytes, container) as ONE fused HIP pipeline.

Differences a maintainer should know (all on the private byte format):
  * the blob is the flat container of include/lmc_format.h, not a pickle;
  * a chunk of any length is a single blob (no 256-token sub-chunks, :301-316);
  * "huggingface" input is read through strides -- no permute copy (:377-378).
There is no CPU path: without the HIP library construction raises.
"""
import threading

import torch

from lmcache_amd import native
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
from lmcache_amd.storage_backend.serde.serde import Serializer
from lmcache_amd.utils import _lmcache_nvtx_annotate

logger = init_logger(__name__)


class CacheGenSerializer(Serializer):
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        native.lib()  # fail loudly at construction if the HIP extension is missing
        self.cachegen_config = CacheGenConfig.from_model_name(metadata.model_name)
        self.chunk_size = config.chunk_size
        self.fmt = metadata.fmt
        self.key_bins = self.cachegen_config.key_bins()
        self.value_bins = self.cachegen_config.value_bins()
        self._lock = threading.Lock()
        self._staging = PinnedArena(slab_bytes=64 << 20)

    def make_key_bins(self, config: CacheGenConfig) -> torch.Tensor:
        """Bins per key layer as the reference holds them: a float32 tensor (cachegen_encoder.py:339-344; on the host
        here -- the kernels take the bins as launch arguments, not as a device tensor)."""
        return torch.tensor(config.key_bins(), dtype=torch.float32)

    def make_value_bins(self, config: CacheGenConfig) -> torch.Tensor:
        """... and per value layer (cachegen_encoder.py:346-350)."""
        return torch.tensor(config.value_bins(), dtype=torch.float32)

    @_lmcache_nvtx_annotate
    def to_bytes(self, tensor: torch.Tensor) -> bytes:
        """[L,2,T,H,D] ("vllm") or [L,2,H,T,D] ("huggingface"), bf16/fp16, any device -> bytes."""
        if self.fmt not in ("vllm", "huggingface"):
            raise ValueError(f"Invalid format: {self.fmt}")
        if not tensor.is_cuda:
            tensor = tensor.cuda()
        with self._lock, torch.cuda.device(tensor.device):
            codec = get_codec(tensor.device.index)
            layout = native.KVLayout.from_chunk(tensor, self.fmt)
            bins = self.cachegen_config.plane_bins(layout.L)
            ntok = layout.ntokens
            job = codec.encode(layout, 0, ntok, ntok, bins)
            sizes = codec.sizes_of(job)
            # bounce through pinned memory, then hand an immutable `bytes` to the caller
            blobs, done = codec.offload(job, sizes, self._staging)
            done.synchronize()
            out = blobs[0].tobytes()
            self._staging.reset()  # single in-flight blob: recycle the slab
            return out
