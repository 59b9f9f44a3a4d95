"""CacheGen per-model quantisation tables and the blob view.

Mirror of lmcache/storage_backend/serde/cachegen_basics.py:
  * CACHEGEN_GPU_MAX_TOKENS_PER_CHUNK (:13) -- kept for API parity only: our
    stream format has no 256-token limit, a chunk of any length is one blob.
  * CacheGenConfig.from_model_name (:17-78) -- same per-family values and the
    same ValueError for unknown models, plus aliases for the model names
    BASELINE.json uses ("Llama-3-8B", "Llama-3-70B": SURVEY.md section 7 edge cases).
  * CacheGenEncoderOutput / CacheGenGPUEncoderOutput (:81-142) -- the reference
    pickles these; here from_bytes() PARSES our flat blob (include/lmc_format.h)
    and exposes the same attribute names (num_heads, head_size, cdf,
    max_tensors_key, max_tensors_value, data_chunks) as read-only views, so
    `CacheGenEncoderOutput.from_bytes(blob).num_heads` (tests/test_serde.py:60-62)
    keeps working without unpickling untrusted bytes.
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from lmcache_amd import native

CACHEGEN_GPU_MAX_TOKENS_PER_CHUNK = 256

_FAMILY_32L = (
    "mistralai/Mistral-7B-Instruct-v0.2", "lmsys/longchat-7b-16k", "Qwen/Qwen-7B",  # family_7b
    "meta-llama/Llama-3.1-8B-Instruct",                                             # family_8b
    # aliases for BASELINE.json's names (same 32-layer table)
    "meta-llama/Meta-Llama-3-8B", "meta-llama/Meta-Llama-3-8B-Instruct", "meta-llama/Llama-3.1-8B",
    "Llama-3-8B", "mistralai/Mistral-7B-v0.1", "Mistral-7B",
)
_FAMILY_40L = ("THUDM/glm-4-9b-chat",)
# 80-layer models (BASELINE config 4).  The reference has no table for them; we keep its
# 10 / 20 / rest key split and 2 / rest value split.
_FAMILY_80L = ("meta-llama/Meta-Llama-3-70B", "meta-llama/Meta-Llama-3-70B-Instruct",
               "meta-llama/Llama-3.1-70B-Instruct", "Llama-3-70B")


@dataclass
class CacheGenConfig:
    key_first_layers: int
    key_second_layers: int
    key_third_layers: int  # = total layers
    key_first_bins: int
    key_second_bins: int
    key_third_bins: int
    value_first_layers: int
    value_first_bins: int
    value_second_bins: int

    def __getitem__(self, key: str) -> int:
        return getattr(self, key)

    @staticmethod
    def from_model_name(model_name: str) -> "CacheGenConfig":
        if model_name in _FAMILY_32L:
            total = 32
        elif model_name in _FAMILY_40L:
            total = 40
        elif model_name in _FAMILY_80L:
            total = 80
        else:
            raise ValueError(f"Model {model_name} is not supported")
        return CacheGenConfig(key_first_layers=10, key_second_layers=20, key_third_layers=total,
                              key_first_bins=32, key_second_bins=16, key_third_bins=16,
                              value_first_layers=2, value_first_bins=32, value_second_bins=16)

    # make_key_bins / make_value_bins of CacheGenSerializer (cachegen_encoder.py:339-350) as host lists
    def key_bins(self) -> List[int]:
        n = self.key_third_layers
        return [self.key_first_bins if l < self.key_first_layers else
                self.key_second_bins if l < self.key_second_layers else self.key_third_bins for l in range(n)]

    def value_bins(self) -> List[int]:
        n = self.key_third_layers
        return [self.value_first_bins if l < self.value_first_layers else self.value_second_bins for l in range(n)]

    def plane_bins(self, num_layers: int) -> List[int]:
        """bins per plane, plane order p = kv*L + layer, for a chunk with `num_layers` layers."""
        kb, vb = self.key_bins(), self.value_bins()
        if num_layers > len(kb):
            raise ValueError(f"chunk has {num_layers} layers but the CacheGen table covers {len(kb)}")
        return kb[:num_layers] + vb[:num_layers]


@dataclass
class CacheGenGPUBytestream:
    bytestream: torch.Tensor          # uint8 1-D: the streams section
    bytestream_lengths: torch.Tensor  # int32 [nplanes, ngroups]: exact bytes per group stream
    ntokens: int


class CacheGenGPUEncoderOutput:
    """Read-only view of one encoded chunk (host memory)."""

    def __init__(self, blob):
        self._blob = bytes(blob) if not isinstance(blob, bytes) else blob
        h = native.blob_info(self._blob)
        self.header = h
        self.num_heads = int(h.num_heads)
        self.head_size = int(h.head_size)
        self.num_layers = int(h.num_layers)
        self.ntokens = int(h.ntokens)
        self.dtype = native.torch_dtype(int(h.dtype))

    def _section(self, off, count, dt):
        return np.frombuffer(self._blob, dtype=dt, count=count, offset=off)

    @property
    def bins(self) -> List[int]:
        return self._section(self.header.off_bins, self.header.nplanes, np.uint8).tolist()

    def _stream_dir(self) -> np.ndarray:
        """[P * G, 2] = {beg, end} of every group stream, relative to the streams section."""
        h = self.header
        return self._section(h.off_gdir, 2 * h.nplanes * h.ngroups, np.uint32).astype(np.int64).reshape(-1, 2)

    def _stream_counts(self, pg: int, R: int) -> np.ndarray:
        """Stored counts [64 lanes, R] of group stream pg, from the bit planes of its head (include/lmc_format.h)."""
        h = self.header
        base = int(h.off_streams) + int(self._stream_dir()[pg, 0])
        R8 = (R + 7) & ~7
        widths = self._section(base, R8, np.uint8)
        if widths[R:].any() or (widths > 16).any():
            raise ValueError("malformed stream head")
        planes = self._section(base + R8, int(widths.sum()), np.uint64)
        bits = ((planes[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
        cnt = np.zeros((64, R), np.int64)
        j = 0
        for i in range(R):
            for _ in range(int(widths[i])):
                cnt[:, i] = (cnt[:, i] << 1) | bits[j]
                j += 1
        return cnt

    @property
    def cdf(self) -> torch.Tensor:
        """int16 [2L, C, 33] -- the reference's `cdf` tensor, rebuilt from the blob: every stream opens with the
        symbol counts of its 64 channels, and cdf[i] = RNE(N_i * 65504 / T) + i with N_i the number of symbols
        < i (cachegen_encoder.py:95-126; include/lmc_format.h)."""
        h = self.header
        P, C, G, LP, T = int(h.nplanes), int(h.nchannels), int(h.ngroups), int(h.lp), int(h.ntokens)
        full = np.empty((P, C, LP), np.uint16)
        i = np.arange(LP, dtype=np.int64)
        for p, b in enumerate(self.bins):
            R = b - 1
            cnt = np.concatenate([self._stream_counts(p * G + g, R) for g in range(G)])[:C]
            if T <= 256:  # a count of 256 reads 255: the counts of a channel sum to T
                short = T - cnt.sum(axis=1)
                rows, cols = np.nonzero((cnt == 255) & (short[:, None] > 0))
                cnt[rows, cols] += short[rows]
            N = np.zeros((C, LP), np.int64)
            N[:, 1:R + 1] = np.cumsum(cnt, axis=1)
            N[:, R + 1:] = N[:, R:R + 1]
            v = N * 65504
            q, r = v // T, v % T
            q = q + ((2 * r > T) | ((2 * r == T) & (q % 2 == 1)))  # round half to even
            full[p] = ((q + i) & 0xffff).astype(np.uint16)
        return torch.from_numpy(full.view(np.int16))

    def _scales(self) -> torch.Tensor:
        h = self.header
        a = self._section(h.off_scales, h.nplanes * h.ntokens, np.int16).reshape(h.nplanes, h.ntokens, 1).copy()
        return torch.from_numpy(a).view(self.dtype)

    @property
    def max_tensors_key(self) -> torch.Tensor:
        return self._scales()[:self.num_layers]

    @property
    def max_tensors_value(self) -> torch.Tensor:
        return self._scales()[self.num_layers:]

    @property
    def data_chunks(self) -> List[CacheGenGPUBytestream]:
        h = self.header
        d = self._stream_dir()
        lens = (d[:, 1] - d[:, 0]).astype(np.int32).reshape(h.nplanes, h.ngroups)
        streams = self._section(h.off_streams, h.stream_bytes, np.uint8).copy()
        return [CacheGenGPUBytestream(torch.from_numpy(streams), torch.from_numpy(lens), self.ntokens)]

    def __getitem__(self, key: str):
        return getattr(self, key)

    def to_bytes(self) -> bytes:
        return self._blob

    @staticmethod
    def from_bytes(bs) -> "CacheGenGPUEncoderOutput":
        return CacheGenGPUEncoderOutput(bs)


# the reference's tests unpickle with this (older) class name (tests/test_serde.py:5,60)
CacheGenEncoderOutput = CacheGenGPUEncoderOutput
