"""LMCLocalBackend -- the local (HBM / host-DRAM) leg of the hot path.

Mirror of the reference's LMCLocalBackend (lmcache/storage_backend/
local_backend.py:22-153): a dict keyed by CacheEngineKey, `put` (blocking or
queued to a worker thread, :72-80, :102-125), `get` returning a tensor on the
GPU (:128-144), `contains`, `close`.

Three storage modes, chosen by the config:
  local_device="cuda"                      chunks stay in HBM (reference :95-100 with device "cuda")
  local_device="cpu"                       raw chunks in PINNED host DRAM: hipMemcpyAsync on a side
                                           stream + events (the reference's pin-memory path is stubbed
                                           off and device-synchronises, :50, :82-90)
  local_device="cpu", local_serde="cachegen"
                                           CacheGen-ENCODED chunks in pinned host DRAM -- the
                                           BASELINE.json north-star path: fused HIP encode, blobs DMA'd
                                           to host on the side stream, ~3x less PCIe and DRAM
  local_device="cuda", local_serde="cachegen"
                                           CacheGen-ENCODED chunks kept in HBM: 4.2x more warm context in the
                                           288 GB than raw chunks, and the one tier whose retrieve (a decode,
                                           no PCIe) can hide behind the model layer by layer
                                           (get_kv_range(..., layers_per_launch=...))

All three implement the optional put_kv_range / get_kv_range protocol
(abstract_backend.py) so the engine never materialises the [L,2,T,H,D] blob
(cache_engine.py:98-161) nor the final torch.cat (:362-368): KV is gathered
from / scattered to the caller's tensors by the HIP kernels.
"""
import ctypes
import os
import queue
import time
import threading
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

from lmcache_amd import native
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.abstract_backend import LMCBackendInterface
from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
from lmcache_amd.storage_backend.serde.cachegen_decoder import output_spec
from lmcache_amd.storage_backend.serde.cachegen_device import DeviceArena, HostBlob, HostPack, PinnedArena, get_codec
from lmcache_amd.utils import CacheEngineKey, _lmcache_nvtx_annotate

logger = init_logger(__name__)


class LocalBackendEndSignal:
    pass


@dataclass
class _DevChunk:
    blob: torch.Tensor                 # uint8 view of the backend's HBM arena: one encoded chunk
    shape: Tuple[int, ...]
    dtype: torch.dtype


@dataclass
class _HostChunk:
    blob: HostBlob
    ready: Optional[torch.cuda.Event]  # D2H finished
    shape: Tuple[int, ...]             # chunk tensor shape in the engine's fmt
    dtype: torch.dtype
    encoded: bool


@dataclass
class _PackChunk:
    """Chunk `index` of a pack: the blobs of one put_kv_range() call, stored plane-major (pack v3) in pinned host DRAM."""
    pack: HostPack
    index: int
    shape: Tuple[int, ...]
    dtype: torch.dtype
    blob: Optional[HostBlob] = None    # the chunk as a blob of its own, reassembled on first single-chunk use
    encoded: bool = True
    ready: Optional[torch.cuda.Event] = None


def _chunk_shape(fmt: str, L: int, T: int, H: int, D: int) -> Tuple[int, ...]:
    if fmt == "vllm":
        return (L, 2, T, H, D)
    if fmt == "huggingface":
        return (L, 2, H, T, D)
    raise ValueError(f"Invalid format: {fmt}")


def _fmt_of_chunk(t: torch.Tensor, fmt_hint: Optional[str]) -> str:
    return fmt_hint or "vllm"


class LMCLocalBackend(LMCBackendInterface):
    supports_kv_layout = True

    def __init__(self, config: LMCacheEngineConfig, metadata: Optional[LMCacheEngineMetadata] = None):
        super().__init__()
        native.lib()  # no CPU fallback: fail at construction if the HIP library is missing
        self.chunk_size = config.chunk_size
        self.config = config
        self.metadata = metadata
        self.fmt = metadata.fmt if metadata is not None else None
        self.device = config.local_device
        self.dst_device = "cuda"
        if self.device == "cuda":
            self.mode = "hbm-cachegen" if config.local_serde == "cachegen" else "hbm"
        elif self.device == "cpu":
            self.mode = "cachegen" if config.local_serde == "cachegen" else "raw"
        else:
            raise ValueError(f"LMCLocalBackend: unsupported local_device {self.device!r}")
        if config.local_serde not in (None, "cachegen"):
            raise ValueError(f"Invalid local_serde: {config.local_serde}")
        self.cachegen_config = None
        if self.mode in ("cachegen", "hbm-cachegen"):
            if metadata is None:
                raise ValueError("local_serde='cachegen' needs the engine metadata (model name, fmt)")
            self.cachegen_config = CacheGenConfig.from_model_name(metadata.model_name)
        self.dict: Dict[CacheEngineKey, Union[torch.Tensor, _HostChunk]] = {}
        self.update_lock = threading.Lock()
        self.host_arena = PinnedArena() if self.mode in ("raw", "cachegen") else None
        # pinned CacheGen tier: a put_kv_range() of several chunks is stored as ONE plane-major pack (lmc_store_pack_parts), so
        # that a retrieve of the same prefix moves a range of layers as one transfer (LMCACHE_AMD_PINNED_PACKS=0: one
        # blob per chunk, as the reference stores them)
        self.pack_stores = self.mode == "cachegen" and os.environ.get("LMCACHE_AMD_PINNED_PACKS", "1") != "0"
        self.dev_arena: Optional[DeviceArena] = None   # hbm-cachegen: created on first use, on the KV's device
        self._stage: Optional[torch.Tensor] = None   # device staging for raw gathers / scatters
        self._stage_free: Optional[torch.cuda.Event] = None
        self._cuda_device = torch.cuda.current_device()
        self.last_publish_time = 0.0
        self._gen = 0               # bumped by every _publish: what _prefix_entries' kept answer is valid for
        self._prefix_memo = None
        self.put_queue: "queue.Queue" = queue.Queue()
        self.put_thread = threading.Thread(target=self.put_worker, daemon=True)
        self.put_thread.start()

    # ------------------------------------------------------------------ basics
    def contains(self, key: CacheEngineKey) -> bool:
        return key in self.dict

    def _prefix_entries(self, keys: Sequence[CacheEngineKey]) -> list:
        """The entries of the leading keys that are present (stop at the first miss).  The last answer is kept for the SAME
        key list object while nothing has been published since (the engine keeps the key lists of its last hash chains:
        the lookup -> retrieve of a warm prefix probes the same 64 keys twice, every call of a serving engine)."""
        m = self._prefix_memo
        if m is not None and m[0] is keys and m[1] == self._gen:
            return m[2]
        d, entries = self.dict, []
        for k in keys:
            e = d.get(k)
            if e is None:
                break
            entries.append(e)
        self._prefix_memo = (keys, self._gen, entries)
        return entries

    def contains_prefix(self, keys: Sequence[CacheEngineKey]) -> int:
        """How many leading keys are present: the engine's prefix probe (cache_engine.py:323-345: stop at the first
        miss) as one call."""
        return len(self._prefix_entries(keys))

    def _publish(self, key: CacheEngineKey, entry) -> None:
        with self.update_lock:
            self.dict[key] = entry
            self._gen += 1
            self.last_publish_time = time.perf_counter()  # (when a non-blocking store became visible: bench.py store_hidden)

    @_lmcache_nvtx_annotate
    def put_worker(self):
        torch.cuda.set_device(self._cuda_device)
        while True:
            item = self.put_queue.get()
            if isinstance(item, LocalBackendEndSignal):
                break
            try:
                item()
            except Exception:  # keep the worker alive; the chunk simply is not cached
                logger.exception("asynchronous put failed")

    def close(self):
        if self.put_thread is not None and self.put_thread.is_alive():
            self.put_queue.put(LocalBackendEndSignal())
            self.put_thread.join()
        if self.host_arena is not None:
            # entries may still be read by a caller holding tensors made from them: drop our refs first
            self.dict.clear()
            self.host_arena.close()
            self.host_arena = None
        if self.dev_arena is not None:
            self.dict.clear()
            self.dev_arena.close()
            self.dev_arena = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ single chunk
    def _codec(self):
        return get_codec(self._cuda_device)

    def _finish_encoded(self, keys: Sequence[CacheEngineKey], job, shapes, dtype) -> None:
        codec = self._codec()
        if self.mode == "hbm-cachegen":
            if self.dev_arena is None:
                self.dev_arena = DeviceArena(torch.device("cuda", self._cuda_device))
            for key, t, shp in zip(keys, codec.keep_on_device(job, self.dev_arena), shapes):
                self._publish(key, _DevChunk(t, shp, dtype))
            return
        blobs, done = codec.offload(job, None, self.host_arena)  # range by range, overlapping the rest of the encode
        done.synchronize()
        for key, hb, shp in zip(keys, blobs, shapes):
            self._publish(key, _HostChunk(hb, None, shp, dtype, True))

    def _finish_pack(self, keys: Sequence[CacheEngineKey], job, shapes, dtype) -> None:
        pack = self._codec().finish_pack(job, self.host_arena)
        for i, (key, shp) in enumerate(zip(keys, shapes)):
            self._publish(key, _PackChunk(pack, i, shp, dtype))

    def _own_blob(self, e: _PackChunk) -> HostBlob:
        """A pack's chunk as a blob of its own in the pinned arena (the single-chunk paths: get(), a retrieve that mixes
        chunks of different stores)."""
        if e.blob is None:
            data = e.pack.extract(e.index)
            hb = self.host_arena.alloc(len(data))
            ctypes.memmove(hb.ptr, data, len(data))
            e.blob = hb
        return e.blob

    def _put_chunk_now(self, key: CacheEngineKey, kv_chunk: torch.Tensor, fmt: str) -> None:
        if not kv_chunk.is_cuda:
            kv_chunk = kv_chunk.to(self.dst_device)
        if self.mode == "hbm":
            self._publish(key, kv_chunk if kv_chunk.is_contiguous() else kv_chunk.contiguous())
            return
        lay = native.KVLayout.from_chunk(kv_chunk, fmt)
        shape = _chunk_shape(fmt, lay.L, lay.ntokens, lay.H, lay.D)
        if self.mode in ("cachegen", "hbm-cachegen"):
            _, out_dt = output_spec(fmt, 1, 1, 1, 8)
            with torch.cuda.device(kv_chunk.device):
                job = self._codec().encode(lay, 0, lay.ntokens, lay.ntokens, self.cachegen_config.plane_bins(lay.L))
            self._finish_encoded([key], job, [shape], out_dt)
            return
        # raw: contiguous chunk -> pinned host
        src = kv_chunk if kv_chunk.is_contiguous() else kv_chunk.contiguous()
        nbytes = src.numel() * src.element_size()
        hb = self.host_arena.alloc(nbytes)
        codec = self._codec()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(src.device))
        codec.copy_stream.wait_event(ev)
        native.memcpy_async(hb.ptr, src.data_ptr(), nbytes, "d2h", codec.copy_stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(codec.copy_stream)
        done.synchronize()  # `src` may be freed by the caller after we return
        self._publish(key, _HostChunk(hb, None, tuple(src.shape), src.dtype, False))

    def put(self, key: CacheEngineKey, kv_chunk: torch.Tensor, blocking: bool = True) -> None:
        fmt = _fmt_of_chunk(kv_chunk, self.fmt)
        if blocking:
            self._put_chunk_now(key, kv_chunk, fmt)
        else:
            # the worker thread queues its work on ITS current stream: order it behind the stream that produced the chunk
            ready = None
            if kv_chunk.is_cuda:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(kv_chunk.device))

            def queued():
                if ready is not None:
                    torch.cuda.current_stream(kv_chunk.device).wait_event(ready)
                self._put_chunk_now(key, kv_chunk, fmt)
            self.put_queue.put(queued)

    @_lmcache_nvtx_annotate
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        entry = self.dict.get(key, None)
        if entry is None:
            return None
        if isinstance(entry, torch.Tensor):
            return entry.to(self.dst_device)
        dev = torch.device("cuda", self._cuda_device)
        out = torch.empty(entry.shape, dtype=entry.dtype, device=dev)
        fmt = self.fmt or "vllm"
        if isinstance(entry, _DevChunk):
            T = entry.shape[2] if fmt == "vllm" else entry.shape[3]
            codec = self._codec()
            try:
                codec.finish_decode(codec.decode_device([entry.blob], native.KVLayout.from_chunk(out, fmt), 0, T))
            except native.NativeError:
                logger.exception("stored chunk does not decode: treated as a miss")
                return None
            return out
        if entry.encoded:
            T = entry.shape[2] if fmt == "vllm" else entry.shape[3]
            codec = self._codec()
            try:
                lay = native.KVLayout.from_chunk(out, fmt)
                if isinstance(entry, _PackChunk) and entry.pack.chunk_tokens >= T:
                    # a chunk of a pack is read where it lies (lmc_load_pack over one chunk): nothing is reassembled on
                    # the host, no pinned memory is taken
                    with torch.cuda.device(dev):
                        codec.finish_decode(codec.load_pack(entry.pack, entry.index, 1, lay, 0, None))
                else:
                    blob = self._own_blob(entry) if isinstance(entry, _PackChunk) else entry.blob
                    codec.finish_decode(codec.decode([blob], lay, 0, T))
            except native.NativeError:
                logger.exception("stored chunk does not decode: treated as a miss")
                return None
        else:
            cur = torch.cuda.current_stream(dev)
            native.memcpy_async(out.data_ptr(), entry.blob.ptr, entry.blob.nbytes, "h2d", cur.cuda_stream)
        return out

    def chunk_meta(self, key: CacheEngineKey) -> Tuple[Tuple[int, ...], torch.dtype]:
        """(shape, dtype) of the chunk tensor get(key) would return."""
        e = self.dict[key]
        return (tuple(e.shape), e.dtype)

    # ------------------------------------------------------------- range protocol
    def _stage_tensor(self, nbytes: int, dev) -> torch.Tensor:
        if self._stage is None or self._stage.numel() < nbytes:
            self._stage = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._stage

    def put_kv_range(self, keys: Sequence[CacheEngineKey], src: native.KVLayout, fmt: str, tok_begin: int,
                     tok_end: int, chunk_tokens: int, blocking: bool = True) -> int:
        """Store chunks [tok_begin + i*chunk_tokens, ...) of `src` under keys[i], reading KV where it lies."""
        n = (tok_end - tok_begin + chunk_tokens - 1) // chunk_tokens
        assert n == len(keys), "one key per chunk"
        if n == 0:
            return 0
        L, H, D = src.L, src.H, src.D
        dt = native.torch_dtype(src.dtype)
        shapes = [_chunk_shape(fmt, L, min(chunk_tokens, tok_end - (tok_begin + i * chunk_tokens)), H, D)
                  for i in range(n)]
        ctx = native.get_context(self._cuda_device)
        dev = src.device
        if self.mode in ("cachegen", "hbm-cachegen"):
            _, out_dt = output_spec(fmt, 1, 1, 1, 8)
            if self.pack_stores and n >= 2:
                with torch.cuda.device(dev):
                    pjob = self._codec().store_pack(src, tok_begin, tok_end, chunk_tokens, self.cachegen_config.plane_bins(L),
                                                    self.host_arena)
                if blocking:
                    self._finish_pack(keys, pjob, shapes, out_dt)
                else:
                    self.put_queue.put(lambda: self._finish_pack(list(keys), pjob, shapes, out_dt))
                return n
            with torch.cuda.device(dev):
                job = self._codec().encode(src, tok_begin, tok_end, chunk_tokens, self.cachegen_config.plane_bins(L))
            if blocking:
                self._finish_encoded(keys, job, shapes, out_dt)
            else:
                self.put_queue.put(lambda: self._finish_encoded(list(keys), job, shapes, out_dt))
            return n
        if self.mode == "hbm":
            for key, shp, i in zip(keys, shapes, range(n)):
                chunk = torch.empty(shp, dtype=dt, device=dev)
                T = shp[2] if fmt == "vllm" else shp[3]
                ctx.copy_kv(src, tok_begin + i * chunk_tokens, T, native.KVLayout.from_chunk(chunk, fmt), 0)
                self._publish(key, chunk)
            return n
        # raw: gather every chunk into a device staging arena with the copy kernel, then D2H on the side stream
        codec = self._codec()
        cur = torch.cuda.current_stream(dev)
        chunk_bytes = L * 2 * chunk_tokens * H * D * 2
        stage = self._stage_tensor(n * chunk_bytes, dev)
        if self._stage_free is not None:
            cur.wait_event(self._stage_free)
        views = []
        for i, shp in enumerate(shapes):
            numel = 1
            for s in shp:
                numel *= s
            v = stage[i * chunk_bytes:i * chunk_bytes + numel * 2].view(dt).view(shp)
            T = shp[2] if fmt == "vllm" else shp[3]
            ctx.copy_kv(src, tok_begin + i * chunk_tokens, T, native.KVLayout.from_chunk(v, fmt), 0)
            views.append(v)
        gathered = torch.cuda.Event()
        gathered.record(cur)
        codec.copy_stream.wait_event(gathered)
        entries = []
        for v in views:
            nb = v.numel() * 2
            hb = self.host_arena.alloc(nb)
            native.memcpy_async(hb.ptr, v.data_ptr(), nb, "d2h", codec.copy_stream.cuda_stream)
            entries.append(_HostChunk(hb, None, tuple(v.shape), dt, False))
        done = torch.cuda.Event()
        done.record(codec.copy_stream)
        self._stage_free = done

        def finish():
            done.synchronize()
            for key, e in zip(keys, entries):
                self._publish(key, e)

        if blocking:
            finish()
        else:
            self.put_queue.put(finish)
        return n

    def get_kv_range(self, keys: Sequence[CacheEngineKey], dst: native.KVLayout, fmt: str, dst_tok0: int,
                     chunk_tokens: int, layers_per_launch: Optional[int] = None, jobs_out: Optional[list] = None) -> int:
        """Write chunk i (stored under keys[i]) to dst tokens dst_tok0 + i*chunk_tokens ...; tokens that land
        below 0 are dropped (retrieve()'s first-chunk trim, cache_engine.py:360-365).  Returns the number of
        leading chunks written (a key that has gone since `contains` ends the run, like the reference's break on
        the first None chunk, cache_engine.py:339-345); raises NativeError if a stored blob does not decode."""
        entries = self._prefix_entries(keys)
        if not entries:
            return 0
        ctx = native.get_context(self._cuda_device)
        dev = dst.device
        if self.mode == "hbm-cachegen":
            # blobs in HBM: one decode launch per range of layers, an event after each; with jobs_out the call
            # returns at once and the caller finishes the job (engine.retrieve_layerwise)
            codec = self._codec()
            with torch.cuda.device(dev):
                # (`entries` is the SAME list object for a repeated lookup of one prefix -- _prefix_entries -- and the codec
                # keeps the uploaded address table of the last few lists it has seen)
                job = codec.decode_device([e.blob for e in entries], dst, dst_tok0, chunk_tokens, layers_per_launch,
                                          same_blobs_as=entries)
            if jobs_out is not None:
                jobs_out.append((codec, job))
            else:
                codec.finish_decode(job)
            return len(entries)
        if self.mode == "cachegen":
            # The entries split into maximal RUNS: consecutive chunks of one pack (what one put_kv_range stored) are one
            # lmc_load_pack -- a transfer and a decode per range of layers, straight from the pack --, a run of chunks
            # with blobs of their own is one decode over those blobs.  A retrieve that spans several stores (every turn
            # of a conversation, every prompt behind a shared prefix adds a pack) is a few such jobs on the same
            # streams, one behind the other; nothing is reassembled on the host and no pinned memory is allocated.
            codec = self._codec()
            runs, i = [], 0
            while i < len(entries):
                e = entries[i]
                j = i + 1
                if isinstance(e, _PackChunk) and e.pack.chunk_tokens == chunk_tokens:
                    while (j < len(entries) and isinstance(entries[j], _PackChunk) and entries[j].pack is e.pack
                           and entries[j].index == e.index + (j - i)):
                        j += 1
                    runs.append(("pack", i, j))
                else:
                    while j < len(entries) and not (isinstance(entries[j], _PackChunk)
                                                    and entries[j].pack.chunk_tokens == chunk_tokens):
                        j += 1
                    runs.append(("blobs", i, j))
                i = j
            layerwise = bool(layers_per_launch) and jobs_out is not None
            for kind, i, j in runs:
                tok0 = dst_tok0 + i * chunk_tokens
                with torch.cuda.device(dev):
                    if kind == "pack":
                        job = codec.load_pack(entries[i].pack, entries[i].index, j - i, dst, tok0, layers_per_launch)
                    else:
                        # (a pack chunk of another chunk length -- never stored by this engine -- takes its blob from the pack)
                        blobs = [self._own_blob(e) if isinstance(e, _PackChunk) else e.blob for e in entries[i:j]]
                        if layerwise:
                            # pinned tier cut by layers (engine.retrieve_layerwise): one lmc_load_chunks call gathers and
                            # decodes range after range; the caller's layers wait for their range's event only
                            job = codec.decode_host_layerwise(blobs, dst, tok0, chunk_tokens, layers_per_launch)
                        else:
                            job = codec.decode(blobs, dst, tok0, chunk_tokens)
                if layerwise:
                    jobs_out.append((codec, job))
                else:
                    codec.finish_decode(job)  # this decode's event, then its own status word
            return len(entries)
        cur = torch.cuda.current_stream(dev)
        stage = None
        if self.mode == "raw":
            tot = sum(native.r16(e.blob.nbytes) for e in entries)
            stage = self._stage_tensor(tot, dev)
            if self._stage_free is not None:
                cur.wait_event(self._stage_free)
        off = 0
        for i, e in enumerate(entries):
            if isinstance(e, torch.Tensor):
                chunk = e
            else:
                numel = e.blob.nbytes // 2
                chunk = stage[off:off + e.blob.nbytes].view(e.dtype)[:numel].view(e.shape)
                native.memcpy_async(chunk.data_ptr(), e.blob.ptr, e.blob.nbytes, "h2d", cur.cuda_stream)
                off += native.r16(e.blob.nbytes)
            T = chunk.shape[2] if fmt == "vllm" else chunk.shape[3]
            t0 = dst_tok0 + i * chunk_tokens
            skip = max(0, -t0)
            if skip < T:
                ctx.copy_kv(native.KVLayout.from_chunk(chunk, fmt), skip, T - skip, dst, t0 + skip)
        if stage is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._stage_free = ev
        return len(entries)


class LMCLocalDiskBackend(LMCBackendInterface):
    """The local-disk tier (mirror of the reference's LMCLocalDiskBackend, lmcache/storage_backend/local_backend.py:
    163-310: `config.local_device` is a directory, one file per chunk named after the key, a set of the keys written
    by this process, `put` blocking or queued to a worker thread, `get` returning a tensor on the GPU).

    Two file formats, chosen like the host tier's by `config.local_serde`:
      None        the reference's own: a safetensors file holding the raw chunk under "kv_chunk" (:246, :300-303) --
                  files written by either implementation are read by the other;
      "cachegen"  the flat v6 blob (include/lmc_format.h) exactly as the GPU wrote it: put = fused HIP encode, DMA into
                  pinned memory, ONE write() from that buffer; get = ONE readinto() a pinned buffer, H2D on the copy
                  stream, decode straight into the returned tensor -- 4.6x less file I/O than the raw chunk, no
                  pickle, no intermediate host copy ("on-disk backend reuse").
    """

    def __init__(self, config: LMCacheEngineConfig, metadata: Optional[LMCacheEngineMetadata] = None):
        super().__init__()
        self.chunk_size = config.chunk_size
        self.config = config
        self.metadata = metadata
        self.path = config.local_device
        assert self.path is not None, "Need to specify local path if when using LMCLocalDiskBackend"
        if config.local_serde not in (None, "cachegen"):
            raise ValueError(f"Invalid local_serde: {config.local_serde}")
        self.encoded = config.local_serde == "cachegen"
        self.cachegen_config = None
        self.fmt = metadata.fmt if metadata is not None else None
        if self.encoded:
            native.lib()  # no CPU fallback: fail at construction if the HIP library is missing
            if metadata is None:
                raise ValueError("local_serde='cachegen' needs the engine metadata (model name, fmt)")
            self.cachegen_config = CacheGenConfig.from_model_name(metadata.model_name)
        if not os.path.exists(self.path):
            os.makedirs(self.path)
        self.existing_keys = set()
        self.update_lock = threading.Lock()
        self._io_lock = threading.Lock()       # one staging buffer per direction
        self._staging = None                   # PinnedArena, created on first encoded put
        self._read_buf: Optional[native.PinnedBuffer] = None
        self.dst_device = "cuda"
        self._cuda_device = torch.cuda.current_device() if torch.cuda.is_available() else None
        self.put_queue: "queue.Queue" = queue.Queue()
        self.put_thread: Optional[threading.Thread] = threading.Thread(target=self.put_worker, daemon=True)
        self.put_thread.start()

    def contains(self, key: CacheEngineKey) -> bool:
        return key in self.existing_keys

    def _key_to_path(self, key: CacheEngineKey) -> str:
        # the reference's naming (:222-236), the extension telling the two formats apart
        return self.path + key.to_string().replace("/", "-") + (".lmc" if self.encoded else ".pt")

    def put_worker(self):
        # the worker's own stream on the ENGINE's device (a thread starts on device 0), ordered behind whatever produced
        # each chunk: put() records an event on the caller's stream -- the engine fills the chunk with copy_kv on that
        # stream and returns at once -- and the queued put waits for it before it reads the tensor
        stream = None
        if self._cuda_device is not None:
            torch.cuda.set_device(self._cuda_device)
            stream = torch.cuda.Stream()
        while True:
            item = self.put_queue.get()
            if isinstance(item, LocalBackendEndSignal):
                break
            key, value, ready = item
            try:
                if stream is not None:
                    with torch.cuda.stream(stream):
                        if ready is not None:
                            stream.wait_event(ready)
                        self.put_blocking(key, value)
                else:
                    self.put_blocking(key, value)
            except Exception:
                logger.exception("queued disk put failed")

    @_lmcache_nvtx_annotate
    def put_blocking(self, key: CacheEngineKey, kv_chunk: torch.Tensor) -> None:
        path = self._key_to_path(key)
        tmp = f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"
        if self.encoded:
            self._write_encoded(kv_chunk, tmp)
        else:
            from safetensors.torch import save_file
            save_file({"kv_chunk": kv_chunk.contiguous()}, tmp)
        os.replace(tmp, path)  # a reader never sees a half-written file
        # the order of "file complete" and "key visible" matters (:247-252)
        with self.update_lock:
            self.existing_keys.add(key)

    def _write_encoded(self, kv_chunk: torch.Tensor, path: str) -> None:
        fmt = _fmt_of_chunk(kv_chunk, self.fmt)
        if not kv_chunk.is_cuda:
            kv_chunk = kv_chunk.to(self.dst_device)
        with self._io_lock, torch.cuda.device(kv_chunk.device):
            if self._staging is None:
                self._staging = PinnedArena(slab_bytes=64 << 20)
            codec = get_codec(kv_chunk.device.index)
            lay = native.KVLayout.from_chunk(kv_chunk, fmt)
            job = codec.encode(lay, 0, lay.ntokens, lay.ntokens, self.cachegen_config.plane_bins(lay.L))
            sizes = codec.sizes_of(job)
            blobs, done = codec.offload(job, sizes, self._staging)
            done.synchronize()
            hb = blobs[0]
            view = (ctypes.c_uint8 * hb.nbytes).from_address(hb.ptr)
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                mv, off = memoryview(view).cast("B"), 0
                while off < hb.nbytes:
                    off += os.write(fd, mv[off:])
            finally:
                os.close(fd)
            self._staging.reset()  # single in-flight blob: recycle the slab

    def put(self, key: CacheEngineKey, kv_chunk: torch.Tensor, blocking: bool = True) -> None:
        if blocking:
            self.put_blocking(key, kv_chunk)
        else:
            ready = None
            if kv_chunk.is_cuda:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(kv_chunk.device))
            self.put_queue.put((key, kv_chunk, ready))

    @_lmcache_nvtx_annotate
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        if key not in self.existing_keys:
            return None
        path = self._key_to_path(key)
        if not self.encoded:
            from safetensors import safe_open
            with safe_open(path, framework="pt", device=self.dst_device) as f:
                return f.get_tensor("kv_chunk")
        fmt = self.fmt or "vllm"
        dev = torch.device("cuda", torch.cuda.current_device())
        with self._io_lock, torch.cuda.device(dev):
            nbytes = os.path.getsize(path)
            if self._read_buf is None or self._read_buf.nbytes < nbytes:
                if self._read_buf is not None:
                    self._read_buf.free()
                self._read_buf = native.PinnedBuffer(max(native.r16(nbytes), 16 << 20))
            view = (ctypes.c_uint8 * nbytes).from_address(self._read_buf.ptr)
            with open(path, "rb", buffering=0) as f:
                mv, off = memoryview(view).cast("B"), 0
                while off < nbytes:
                    n = f.readinto(mv[off:])
                    if not n:
                        break
                    off += n
            try:
                if off != nbytes:
                    raise native.NativeError(f"short read: {off} of {nbytes} bytes")
                h = native.blob_info(bytes(mv[:native.HEADER_BYTES]), nbytes)
                shape, dtype = output_spec(fmt, h.num_layers, h.ntokens, h.num_heads, h.head_size)
                out = torch.empty(shape, dtype=dtype, device=dev)
                codec = get_codec(dev.index)
                hb = HostBlob(self._read_buf, 0, nbytes)
                # finish_decode waits for this decode: the read buffer is free again when the lock is released
                codec.finish_decode(codec.decode([hb], native.KVLayout.from_chunk(out, fmt), 0, h.ntokens))
            except native.NativeError:
                logger.exception("stored chunk does not decode: treated as a miss")
                return None
            return out

    def close(self):
        if self.put_thread is not None and self.put_thread.is_alive():
            self.put_queue.put(LocalBackendEndSignal())
            self.put_thread.join()
            logger.info("Closed the put worker in local disk backend")
        self.put_thread = None
        with self._io_lock:
            if self._staging is not None:
                self._staging.close()
                self._staging = None
            if self._read_buf is not None:
                self._read_buf.free()
                self._read_buf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
