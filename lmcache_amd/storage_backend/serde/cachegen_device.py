"""Device-side plumbing shared by the CacheGen serializer, deserializer and the
pinned-host backend: HBM blob arena, pinned host arena, side copy stream,
stream/event ordering.  All compute goes through the C ABI (lmcache_amd.native);
this file only sequences it.

What it replaces in the reference: the implicit, synchronous data movement of
`pickle.dump` of CUDA tensors (cachegen_basics.py:131-136), `.cuda()` after
`pickle.load` (cachegen_decoder.py:144-147) and the pageable `.to("cpu")` /
`.to("cuda")` copies of LMCLocalBackend (local_backend.py:82-100, 128-144) --
with pinned `hipMemcpyAsync` on a side stream ordered by events (no device-wide
synchronisation: the reference itself flags torch.cuda.synchronize() as harmful
here, local_backend.py:83-85).
"""
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from lmcache_amd import native
from lmcache_amd.logging import init_logger

logger = init_logger(__name__)


class PinnedArena:
    """Bump allocator over hipHostMalloc'ed slabs.  The reference never evicts
    (hybrid_backend.py:24), so neither do we: memory is returned at close()."""

    def __init__(self, slab_bytes: int = 256 << 20):
        self.slab_bytes = slab_bytes
        self._slabs: List[native.PinnedBuffer] = []
        self._spare: List[native.PinnedBuffer] = []   # slabs allocated ahead of need (reserve)
        self._used = 0
        self._lock = threading.Lock()
        self.total_allocated = 0

    def alloc(self, nbytes: int, slab_hint: int = 0) -> "HostBlob":
        """slab_hint: size of the slab to allocate if a new one is needed (a pack is allocated at its bound and cut
        to its size afterwards: a slab of a few bounds keeps the cut-off tails usable)."""
        need = native.r16(max(nbytes, 16))
        with self._lock:
            if not self._slabs or self._used + need > self._slabs[-1].nbytes:
                k = next((i for i, b in enumerate(self._spare) if b.nbytes >= need), None)
                self._slabs.append(self._spare.pop(k) if k is not None
                                   else native.PinnedBuffer(max(self.slab_bytes, need, slab_hint)))
                self._used = 0
            slab = self._slabs[-1]
            off = self._used
            self._used += need
            self.total_allocated += need
        return HostBlob(slab, off, nbytes)

    def shrink(self, hb: "HostBlob", nbytes: int) -> "HostBlob":
        """Give back the tail of `hb` if it is still the arena's last allocation (a pack is allocated at its bound and
        cut to its size once the GPU has written it); otherwise the tail stays unused."""
        keep = native.r16(max(nbytes, 16))
        with self._lock:
            if self._slabs and hb.slab is self._slabs[-1] and hb.offset + native.r16(max(hb.nbytes, 16)) == self._used:
                self.total_allocated -= self._used - (hb.offset + keep)
                self._used = hb.offset + keep
        return HostBlob(hb.slab, hb.offset, nbytes)

    def reserve(self, nbytes: int, slab_bytes: int = 0) -> None:
        """Allocate slabs for `nbytes` more bytes now (hipHostMalloc of hundreds of MB takes tens of ms and stalls
        the device: a backend sized for its working set pays that at start-up, not inside a store)."""
        slab = max(self.slab_bytes, slab_bytes)
        with self._lock:
            have = sum(b.nbytes for b in self._spare)
            while have < nbytes:
                self._spare.append(native.PinnedBuffer(slab))
                have += slab

    def reset(self):
        """Recycle the newest slab (callers that own every blob handed out so far)."""
        with self._lock:
            for s in self._slabs[:-1]:
                s.free()
            self._slabs = self._slabs[-1:]
            self._used = 0
            self.total_allocated = 0

    def close(self):
        with self._lock:
            for s in self._slabs + self._spare:
                s.free()
            self._slabs, self._spare = [], []


@dataclass
class HostBlob:
    """One encoded chunk resident in pinned host DRAM."""
    slab: native.PinnedBuffer
    offset: int
    nbytes: int

    @property
    def ptr(self) -> int:
        return self.slab.ptr + self.offset

    def tobytes(self) -> bytes:
        return ctypes.string_at(self.ptr, self.nbytes)


@dataclass
class EncodeJob:
    nchunks: int
    stride: int
    arena: torch.Tensor        # device uint8, blob i at i*stride
    sizes: Optional[native.PinnedBuffer]  # this job's own uint32 [nchunks], written by the GPU; back in the pool once read
    done: torch.cuda.Event     # recorded after the last encode kernel
    geometry: tuple            # (L, H, D, chunk_tokens)
    status_idx: int = -1       # this job's status word (native.StatusWords); -1 once read
    size_list: Optional[List[int]] = None  # filled by sizes_of / offload
    # a long job is launched as a few consecutive ranges of chunks, each with its own event, so that the
    # host-DRAM offload of range r can start while range r+1 is still being encoded: (chunk0, chunk1, event)
    parts: Optional[list] = None
    offload_issued: bool = False  # its device -> host copies are on the copy streams (the arena may be reused after them)
    pool: Optional[native.StatusWords] = None  # where status_idx goes back to if nobody reads it (see __del__)

    def __del__(self):  # a job dropped unread (an exception between launch and completion): its status word returns
        if _return_status_word is not None:  # (module globals are gone at interpreter shutdown)
            _return_status_word(self)


@dataclass
class HostPack:
    """The blobs of one store call in pinned host DRAM, laid out plane-major (include/lmc_format.h, "pack" v3)."""
    blob: HostBlob
    nchunks: int
    chunk_tokens: int

    def extract(self, chunk: int) -> bytes:
        """Chunk `chunk` as the blob lmc_encode_chunks wrote (host-side reassembly)."""
        return native.pack_extract(self.blob.ptr, self.blob.nbytes, chunk)


@dataclass
class PackJob:
    """One store_pack() in flight."""
    region: HostBlob           # where the GPU writes the pack (allocated at an upper bound)
    nchunks: int
    chunk_tokens: int
    sizes: Optional[native.PinnedBuffer]
    done: torch.cuda.Event
    status_idx: int = -1
    pool: Optional[native.StatusWords] = None
    dev: Optional[torch.Tensor] = None   # dma=True: the HBM region the pack is written to first
    d2h_issued: bool = True              # ... and whether its copies to pinned memory have been queued
    # dma=True: the encode went out in plane ranges (lmc_store_pack_parts); part r's bytes may leave once part_events[r]
    # has fired: part_info (pinned uint64 [2 nparts]) says where they lie
    part_events: Optional[list] = None
    part_info: Optional[native.PinnedBuffer] = None
    cap: int = 0
    geometry: tuple = ()                 # (L, H, D)

    def __del__(self):
        if _return_status_word is not None:
            _return_status_word(self)


def pack_cap(n: int, L: int, T: int, H: int, D: int, bins: Sequence[int]) -> int:
    """Bytes a pack of n T-token chunks can need: the static sections plus, per group stream, its largest head (every
    symbol of the plane at the widest count) and the coder's bound -- a lane emits at most T * log2(symbols) + 48 bits
    (DESIGN.md "stream bound"; chunks below 256 tokens: + 1.44 bit per symbol kind for the rounding of the scaled model
    counts, lmc_format.h) -- with 3 % on top."""
    import math
    G = (H * D + native.LANES - 1) // native.LANES
    static = native.r16(native.blob_static_bytes(L, T, H, D))
    width = (255 if T == 256 else T).bit_length()
    streams = 0
    for b in bins:
        lane_bytes = math.ceil((T * math.log2(max(2, b - 1)) * 1.03 + 128) / 8) + 4
        head = native.r16(((b - 1 + 7) & ~7) + 8 * (b - 1) * width)
        streams += G * (head + native.r16(native.LANES * lane_bytes))
    return min(native.pack_bound(n, L, T, H, D), native.r16(256 + 8 * (2 * L * n + 1)) + n * (static + streams))


def layer_ranges(L: int, layers_per_launch) -> list:
    """[(first layer, end layer), ...] covering 0..L: one range when layers_per_launch is None / 0, ranges of a
    fixed size for an int, and for a sequence of sizes the ranges it lists, its last size repeating."""
    if not layers_per_launch:
        return [(0, L)]
    sizes = [layers_per_launch] if isinstance(layers_per_launch, int) else list(layers_per_launch)
    out, l0, i = [], 0, 0
    while l0 < L:
        step = max(1, min(L - l0, int(sizes[min(i, len(sizes) - 1)])))
        out.append((l0, l0 + step))
        l0 += step
        i += 1
    return out


@dataclass
class DecodeJob:
    """One decode() call in flight: `done` fires after its last kernel, `status_idx` is its own status word.
    layer_events (decode_device with layers_per_launch): (first layer after the range, event) per launch, in layer
    order -- the KV of layers below `first layer after` is complete once the event has fired."""
    done: torch.cuda.Event
    status_idx: int
    layer_events: Optional[list] = None
    pool: Optional[native.StatusWords] = None

    def __del__(self):  # a DecodeJob / LayerwiseRetrieval nobody finished
        if _return_status_word is not None:
            _return_status_word(self)


def _return_status_word(job) -> None:
    """Give an unread status word back to its pool once the job's kernels can no longer write it."""
    idx, pool = getattr(job, "status_idx", -1), getattr(job, "pool", None)
    if pool is None or idx is None or idx < 0:
        return
    try:
        job.done.synchronize()
    except Exception:
        pass
    try:
        pool.read_release(idx)
    except Exception:
        pass
    job.status_idx = -1


class DeviceArena:
    """Bump allocator over HBM slabs for encoded chunks that stay on the GPU (LMCLocalBackend, local_device="cuda" +
    local_serde="cachegen": 4.2x more warm context in the 288 GB than raw chunks).  No eviction, like the
    reference (hybrid_backend.py:24)."""

    def __init__(self, device, slab_bytes: int = 512 << 20):
        self.device, self.slab_bytes = device, slab_bytes
        self._slabs: List[torch.Tensor] = []
        self._used = 0
        self._lock = threading.Lock()

    def alloc(self, nbytes: int) -> torch.Tensor:
        need = native.r16(max(nbytes, 16))
        with self._lock:
            if not self._slabs or self._used + need > self._slabs[-1].numel():
                self._slabs.append(torch.empty(max(self.slab_bytes, need), dtype=torch.uint8, device=self.device))
                self._used = 0
            off = self._used
            self._used += need
            return self._slabs[-1][off:off + nbytes]

    def close(self):
        with self._lock:
            self._slabs = []


class CacheGenDeviceCodec:
    """Per-device encode/decode sequencer."""

    def __init__(self, device: Optional[int] = None):
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.ctx = native.get_context(self.device_index)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # a second DMA queue for the device -> host leg: two hipMemcpyAsync streams keep two SDMA engines busy
        self.copy_stream2 = torch.cuda.Stream(device=self.device)
        self.d2h_streams = 2
        self.encode_parts = 4                                # ranges a long encode job is launched in
        self._lock = threading.RLock()
        self._enc_arena: Optional[torch.Tensor] = None
        self._dec_arena: Optional[torch.Tensor] = None
        self._size_pool: List[native.PinnedBuffer] = []      # pinned size words, one buffer per job in flight
        self._meta_pool: List[native.PinnedBuffer] = []      # pinned pointer / size arrays of decode_host_layerwise jobs
        self._status = native.StatusWords()                  # one status word per job in flight
        self._arena_free: Optional[torch.cuda.Event] = None  # D2H of the last job that used the shared arena done
        self._dec_free: Optional[torch.cuda.Event] = None    # previous decode kernel done
        self._stage: Optional[native.PinnedBuffer] = None    # staging for pageable `bytes` inputs
        self._shared_job: Optional[EncodeJob] = None         # last job that encoded into the shared arena
        self.decode_batch_chunks = 8                         # chunks per H2D/decode pipeline stage
        self._pack_dev: Optional[torch.Tensor] = None        # HBM staging of a pack on its way to pinned memory (store_pack)
        self._pack_dev_free: Optional[torch.cuda.Event] = None
        self._pack_prev: Optional[PackJob] = None
        self._hdr: Optional[native.PinnedBuffer] = None
        # plane ranges a pack's encode is launched in (store_pack, dma): LMCACHE_AMD_PACK_PARTS=1 is the round-5 behaviour
        # (the whole encode, then the pack, then its copies), for A/B
        self.pack_parts = max(1, min(16, int(os.environ.get("LMCACHE_AMD_PACK_PARTS", "8"))))
        self._part_info_pool: List[native.PinnedBuffer] = []
        self._same_blobs: dict = {}                          # id(caller's list) -> (the list, its blobs' address tuple)
        self._table_cache: dict = {}                         # blob-address tuple -> (device table, largest blob, upload stream)

    # ---- encode ------------------------------------------------------------------
    def _readable(self, src: native.KVLayout, tok_begin: int, tok_end: int):
        """The encoders read 16-byte vectors (native.KVLayout.vector_readable).  The reference's serde takes any shape
        (torch_quant_vectorized, cachegen_encoder.py:40-61), so a range of a layout that is not -- a head_size that is no
        multiple of 8 under a huggingface / NHBD layout, rows off a 16-byte boundary -- is first brought into a
        contiguous vllm chunk on the device (lmc_copy_kv copies element-wise then) and encoded from there."""
        if src.vector_readable():
            return src, tok_begin, tok_end
        n = tok_end - tok_begin
        with torch.cuda.device(self.device):
            chunk = torch.empty((src.L, 2, n, src.H, src.D), dtype=native.torch_dtype(src.dtype), device=src.device)
            self.ctx.copy_kv(src, tok_begin, n, native.KVLayout.from_chunk(chunk, "vllm"), 0)
        return native.KVLayout.from_chunk(chunk, "vllm"), 0, n

    def encode(self, src: native.KVLayout, tok_begin: int, tok_end: int, chunk_tokens: int,
               bins: Sequence[int]) -> EncodeJob:
        """Launch the fused encode of every chunk of [tok_begin, tok_end) on the CURRENT stream
        (so it is ordered after whatever produced the KV).  Asynchronous."""
        src, tok_begin, tok_end = self._readable(src, tok_begin, tok_end)
        L, H, D = src.L, src.H, src.D
        n = (tok_end - tok_begin + chunk_tokens - 1) // chunk_tokens
        stride = native.r16(native.blob_bound(L, chunk_tokens, H, D))
        with self._lock:
            with torch.cuda.device(self.device):
                # The shared arena may be rewritten only after the copies of the last job that used it have been
                # ISSUED (then `_arena_free` orders us behind them).  A non-blocking store defers its offload to the
                # backend's worker thread: while that has not happened, a new job gets an arena of its own.
                prev = self._shared_job
                if prev is None or prev.offload_issued:
                    if self._enc_arena is None or self._enc_arena.numel() < n * stride:
                        self._enc_arena = torch.empty(n * stride, dtype=torch.uint8, device=self.device)
                    arena = self._enc_arena
                else:
                    arena = torch.empty(n * stride, dtype=torch.uint8, device=self.device)
                # every job owns its size words and its status word: a second store never waits for the first
                # one's sizes to be read (the reference's own note on this path: "synchronize is harmful",
                # local_backend.py:83-90), and never sees its failures
                sizes = None
                for k, b in enumerate(self._size_pool):
                    if b.nbytes >= 4 * n:
                        sizes = self._size_pool.pop(k)
                        break
                if sizes is None:
                    sizes = native.PinnedBuffer(4 * max(n, 256))
                st = self._status.acquire()
                cur = torch.cuda.current_stream(self.device)
                try:
                    if arena is self._enc_arena and self._arena_free is not None:
                        cur.wait_event(self._arena_free)  # previous job's D2H has read the arena
                    nparts = self.encode_parts if n >= 4 * self.encode_parts else 1
                    per = (n + nparts - 1) // nparts
                    parts = []
                    for c0 in range(0, n, per):
                        c1 = min(n, c0 + per)
                        self.ctx.encode_chunks(src, tok_begin + c0 * chunk_tokens, min(tok_end, tok_begin + c1 * chunk_tokens),
                                               chunk_tokens, bins, arena.data_ptr() + c0 * stride, stride,
                                               sizes.ptr + 4 * c0, stream=cur.cuda_stream, status_ptr=self._status.ptr(st))
                        ev = torch.cuda.Event()
                        ev.record(cur)
                        parts.append((c0, c1, ev))
                    done = parts[-1][2]
                except BaseException:
                    self._abandon_status(st, cur)
                    self._size_pool.append(sizes)
                    raise
            job = EncodeJob(n, stride, arena, sizes, done, (L, H, D, chunk_tokens), st, None, parts, pool=self._status)
            if arena is self._enc_arena:
                self._shared_job = job
            return job

    def _abandon_status(self, st: int, stream) -> None:
        """A launch sequence failed half way: whatever was queued may still write the word, so it goes back to the
        pool only after the stream has drained (an error path: the host wait does not matter)."""
        try:
            stream.synchronize()
        except Exception:
            pass
        self._status.read_release(st)

    def _check_job(self, job: EncodeJob) -> None:
        """The job's kernels have completed: read its status word (once) and raise on a device error."""
        if job.status_idx >= 0:
            st = self._status.read_release(job.status_idx)
            job.status_idx = -1
            if st:
                raise native.NativeError("CacheGen encode: " + native.describe_status(st))

    def _release_sizes(self, job: EncodeJob) -> None:
        if job.sizes is not None:
            self._size_pool.append(job.sizes)
            job.sizes = None

    def sizes_of(self, job: EncodeJob) -> List[int]:
        """Wait for THIS job only (event, not device) and read the blob sizes the GPU wrote to pinned memory."""
        with self._lock:
            if job.size_list is None:
                job.done.synchronize()
                job.size_list = job.sizes.tensor[:4 * job.nchunks].view(torch.int32).tolist()
                self._release_sizes(job)
                self._check_job(job)
            return job.size_list

    def offload(self, job: EncodeJob, sizes: Optional[Sequence[int]], arena: PinnedArena) -> (List[HostBlob], torch.cuda.Event):
        """hipMemcpyAsync every blob to pinned host DRAM on the side streams, exact sizes.  With sizes=None the
        sizes are read range by range as the job's ranges complete, so the copies of one range overlap the
        encode of the next (job.size_list is filled on the way)."""
        blobs = []
        with self._lock:
            streams = [self.copy_stream]
            if self.d2h_streams > 1 and job.nchunks > 1:
                streams.append(self.copy_stream2)
            progressive = sizes is None and job.size_list is None and job.parts is not None
            if progressive:
                sizes = []
            elif sizes is None:
                sizes = self.sizes_of(job)
            for c0, c1, ev in (job.parts if progressive else [(0, job.nchunks, job.done)]):
                if progressive:
                    ev.synchronize()  # this range only
                    sizes.extend(job.sizes.tensor[4 * c0:4 * c1].view(torch.int32).tolist())
                for st in streams:
                    st.wait_event(ev)
                for i in range(c0, c1):
                    hb = arena.alloc(sizes[i])
                    native.memcpy_async(hb.ptr, job.arena.data_ptr() + i * job.stride, sizes[i], "d2h",
                                        streams[i % len(streams)].cuda_stream)
                    blobs.append(hb)
            if progressive:
                job.size_list = list(sizes)
                self._release_sizes(job)
                self._check_job(job)  # every range's event has fired
            if len(streams) > 1:  # fold the second queue into the first: one event covers both
                ev2 = torch.cuda.Event()
                ev2.record(self.copy_stream2)
                self.copy_stream.wait_event(ev2)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            job.offload_issued = True
            if job.arena is self._enc_arena:
                self._arena_free = ev
        return blobs, ev

    def keep_on_device(self, job: EncodeJob, arena: DeviceArena) -> List[torch.Tensor]:
        """The job's blobs copied (exact sizes, device to device, on the current stream) into a persistent HBM arena
        -- the store leg of the HBM-resident CacheGen tier."""
        sizes = self.sizes_of(job)
        out = []
        with self._lock, torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(job.done)
            for i, n in enumerate(sizes):
                t = arena.alloc(n)
                native.memcpy_async(t.data_ptr(), job.arena.data_ptr() + i * job.stride, n, "d2d", cur.cuda_stream)
                out.append(t)
            ev = torch.cuda.Event()
            ev.record(cur)
            job.offload_issued = True
            if job.arena is self._enc_arena:
                self._arena_free = ev
        return out

    # ---- decode ------------------------------------------------------------------
    def decode_device(self, blobs: Sequence[torch.Tensor], dst: native.KVLayout, dst_tok0: int, chunk_tokens: int,
                      layers_per_launch: Optional[int] = None, same_blobs_as=None) -> Optional[DecodeJob]:
        """Decode blobs that live in HBM (uint8 CUDA tensors, anywhere) straight into `dst` on the current stream,
        no staging copy: the kernel takes the blob addresses from a pointer table.  With layers_per_launch the
        retrieve is cut into one launch per range of layers with an event after each (DecodeJob.layer_events): the
        model can start on layer 0 after 1/L of the decode.  layers_per_launch is a range size or a schedule of
        range sizes whose last entry repeats, e.g. (2, 2, 4, 8, 16): small ranges first so that the model starts
        early, large ones later (a launch of few layers does not fill the GPU, and every launch costs an event)."""
        n = len(blobs)
        if n == 0:
            return None
        L = dst.L
        ranges = layer_ranges(L, layers_per_launch)
        with self._lock, torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            # The blob addresses of THIS call: any number of chunks (a 128 k-token retrieve is 512 of them), uploaded
            # stream-ordered from pinned memory of torch's caching host allocator -- no host wait.  The last few tables are
            # kept, keyed by the addresses themselves (round 6): the warm prefix of a serving engine -- a system prompt, the
            # earlier turns of a conversation -- is the SAME blob set call after call, and building + uploading the table was
            # a third of the host time in front of the first decode launch.  (A table is a function of its key: a stale
            # entry cannot be wrong, only unused.)
            # same_blobs_as: an object the caller hands over again whenever -- and only when -- `blobs` are the same
            # tensors (the backend's kept entry list of a prefix): then not even the 64 data_ptr() calls are repeated.  The
            # object is kept alive beside its key, so its id cannot be recycled while the entry exists.
            ptrs = None
            if same_blobs_as is not None:
                known = self._same_blobs.get(id(same_blobs_as))
                if known is not None and known[0] is same_blobs_as:
                    ptrs = known[1]
            if ptrs is None:
                ptrs = tuple(b.data_ptr() for b in blobs)
                if same_blobs_as is not None:
                    if len(self._same_blobs) >= 8:
                        self._same_blobs.pop(next(iter(self._same_blobs)))
                    self._same_blobs[id(same_blobs_as)] = (same_blobs_as, ptrs)
            hit = self._table_cache.get(ptrs)
            if hit is None:
                table = torch.tensor(ptrs, dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)
                bound = max(b.numel() for b in blobs)
                if len(self._table_cache) >= 8:
                    self._table_cache.pop(next(iter(self._table_cache)))
                self._table_cache[ptrs] = (table, bound, cur)
            else:
                table, bound, up = hit
                if up is not cur:  # uploaded on another stream: order this one behind that copy (once)
                    cur.wait_stream(up)
                    self._table_cache[ptrs] = (table, bound, cur)
            st = self._status.acquire()
            try:
                # ONE C-ABI call issues every range's launch and records its event (lmc_decode_chunks_schedule): the
                # ranges used to be one ctypes call + one torch event each
                ends = [l1 for _, l1 in ranges]
                evs = [native.NativeEvent() for _ in ranges]
                self.ctx.decode_chunks_schedule(table.data_ptr(), bound, n, dst, dst_tok0, chunk_tokens, ends, evs,
                                                stream=cur.cuda_stream, status_ptr=self._status.ptr(st))
                events = list(zip(ends, evs))
            except BaseException:
                self._abandon_status(st, cur)
                raise
            job = DecodeJob(events[-1][1], st, events if layers_per_launch else None, pool=self._status)
            job._table = table  # the kernels read it: alive as long as the job
            return job

    def decode_host_layerwise(self, host_blobs: Sequence["HostBlob"], dst: native.KVLayout, dst_tok0: int,
                              chunk_tokens: int, layers_per_range) -> Optional[DecodeJob]:
        """Blobs in pinned host DRAM -> decoded KV, cut by layers through ONE C-ABI call (lmc_load_chunks): a gather
        kernel pulls the bytes of a range of layers over PCIe while the previous range is decoded, an event per
        range (DecodeJob.layer_events) lets the model start on layer 0 after 1/L of the transfer -- where decode()
        moves whole chunks first (the first layer is complete when the last chunk has landed).
        layers_per_range: an int (a schedule is not supported by the single call: its first entry is used)."""
        n = len(host_blobs)
        if n == 0:
            return None
        step = layers_per_range if isinstance(layers_per_range, int) else int(list(layers_per_range)[0])
        step = max(1, min(dst.L, int(step or dst.L)))
        ranges = layer_ranges(dst.L, step)
        with self._lock, torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            meta = None
            for k, b in enumerate(self._meta_pool):
                if b.nbytes >= 12 * n:
                    meta = self._meta_pool.pop(k)
                    break
            if meta is None:
                meta = native.PinnedBuffer(12 * max(n, 256))
            meta.tensor[:8 * n].view(torch.int64).copy_(torch.tensor([hb.ptr for hb in host_blobs], dtype=torch.int64))
            meta.tensor[8 * n:12 * n].view(torch.int32).copy_(torch.tensor([hb.nbytes for hb in host_blobs], dtype=torch.int32))
            events = [native.NativeEvent() for _ in ranges]
            handles = (ctypes.c_void_p * len(events))(*[e.handle for e in events])
            st = self._status.acquire()
            try:
                self.ctx.load_chunks(meta.ptr, meta.ptr + 8 * n, n, dst, dst_tok0, chunk_tokens, step,
                                     ctypes.cast(handles, ctypes.c_void_p).value, stream=cur.cuda_stream,
                                     status_ptr=self._status.ptr(st))
            except BaseException:
                self._abandon_status(st, cur)
                self._meta_pool.append(meta)
                raise
            job = DecodeJob(events[-1], st, [(l1, ev) for (_, l1), ev in zip(ranges, events)], pool=self._status)
            job._meta, job._meta_pool = meta, self._meta_pool  # the kernels read the arrays: back in the pool at finish
            return job

    # ---- packs: the plane-major pinned tier ---------------------------------------------------------------
    def store_pack(self, src: native.KVLayout, tok_begin: int, tok_end: int, chunk_tokens: int, bins: Sequence[int],
                   arena: PinnedArena, dma: bool = True) -> PackJob:
        """lmc_store_pack on the CURRENT stream: encode every chunk of [tok_begin, tok_end), then a copy kernel writes
        the blobs transposed (static sections, then streams ordered layer / K,V / chunk) into one region.  No host
        wait here; finish_pack() returns the pack in pinned host DRAM.
        dma=True (default): the region is in HBM and finish_pack() moves the finished pack with two DMA copies of its
        exact size -- the copy kernel then runs at HBM speed (0.3 ms) and the PCIe leg disturbs nobody.
        dma=False: the region IS the pinned arena and the copy kernel's stores cross PCIe themselves: one call, no host
        wait at all, but kernels that run beside 11 ms of shader stores to host memory were measured 4.3x slower
        (bench.py store_hidden), so this is for callers with an otherwise idle GPU."""
        src, tok_begin, tok_end = self._readable(src, tok_begin, tok_end)
        L, H, D = src.L, src.H, src.D
        n = (tok_end - tok_begin + chunk_tokens - 1) // chunk_tokens
        with self._lock, torch.cuda.device(self.device):
            cap = pack_cap(n, L, chunk_tokens, H, D, bins)
            cur = torch.cuda.current_stream(self.device)
            region = dev = None
            if dma:
                prev = self._pack_prev
                if prev is None or prev.d2h_issued:
                    if self._pack_dev is None or self._pack_dev.numel() < cap:
                        self._pack_dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
                    dev = self._pack_dev
                    if self._pack_dev_free is not None:
                        cur.wait_event(self._pack_dev_free)  # the previous pack has left the buffer
                else:
                    dev = torch.empty(cap, dtype=torch.uint8, device=self.device)  # the previous store has not been finished yet
            else:
                region = arena.alloc(cap, slab_hint=min(4 * cap, 4 << 30))
            sizes = None
            for k, b in enumerate(self._size_pool):
                if b.nbytes >= 4 * n:
                    sizes = self._size_pool.pop(k)
                    break
            if sizes is None:
                sizes = native.PinnedBuffer(4 * max(n, 256))
            st = self._status.acquire()
            part_events = part_info = None
            try:
                if dma:
                    # the encode in plane ranges, each packed as soon as it is coded (lmc_store_pack_parts): finish_pack
                    # sends a range over PCIe while the later planes are still being encoded
                    nparts = self.pack_parts if n * 2 * L >= 16 * self.pack_parts else 1
                    part_events = [native.NativeEvent() for _ in range(nparts)]
                    part_info = self._part_info_pool.pop() if self._part_info_pool else native.PinnedBuffer(16 * 16)
                    self.ctx.store_pack_parts(src, tok_begin, tok_end, chunk_tokens, bins, dev.data_ptr(), cap, sizes.ptr,
                                              nparts, part_info.ptr, part_events, stream=cur.cuda_stream,
                                              status_ptr=self._status.ptr(st))
                else:
                    self.ctx.store_pack(src, tok_begin, tok_end, chunk_tokens, bins, region.ptr, cap,
                                        sizes.ptr, stream=cur.cuda_stream, status_ptr=self._status.ptr(st))
                done = torch.cuda.Event()
                done.record(cur)
            except BaseException:
                self._abandon_status(st, cur)
                self._size_pool.append(sizes)
                if part_info is not None:
                    self._part_info_pool.append(part_info)
                raise
            job = PackJob(region, n, chunk_tokens, sizes, done, st, pool=self._status)
            job.dev, job.d2h_issued = dev, not dma
            job.geometry = (L, H, D)
            job.part_events, job.part_info, job.cap = part_events, part_info, cap
            if dma and dev is self._pack_dev:
                self._pack_prev = job
            return job

    def finish_pack(self, job: PackJob, arena: PinnedArena) -> HostPack:
        """Wait for THIS store (its event), raise NativeError if a kernel flagged it, return the pack in pinned host DRAM
        (dma=True: its size is read from the header the GPU wrote, the pinned region is allocated at that size and two
        DMA queues move one half each)."""
        if job.dev is None:
            job.done.synchronize()
            st = self._release_pack_words(job)
            if st:
                raise native.NativeError("CacheGen store (pack): " + native.describe_status(st))
            h = native.pack_info(job.region.ptr, job.region.nbytes)
            return HostPack(arena.shrink(job.region, h.total_bytes), job.nchunks, job.chunk_tokens)
        # dma: the pack is being built in HBM part by part.  Part r leaves as soon as its event has fired -- the host waits
        # for that event only, reads where the part lies (two pinned words the GPU wrote) and queues ONE DMA copy, on
        # alternating copy streams (two DMA queues) -- while the later plane ranges are still being encoded; behind the
        # last part the static sections (header, offset table, static slots) follow.  The pinned region is taken at the
        # pack's upper bound and cut to its size afterwards.
        off_streams = native.pack_off_streams(job.nchunks, job.geometry[0], job.chunk_tokens, job.geometry[1], job.geometry[2])
        region = arena.alloc(job.cap, slab_hint=min(4 * job.cap, 4 << 30))
        info = job.part_info.tensor.view(torch.int64)
        streams = [self.copy_stream, self.copy_stream2]
        total, failed = off_streams, False
        with torch.cuda.device(self.device):
            for r, ev in enumerate(job.part_events):
                ev.synchronize()  # this part only
                off, nbytes = int(info[2 * r]), int(info[2 * r + 1])
                if nbytes <= 0:
                    failed = failed or (r == 0)  # (parts behind the only one of an unsplit job are empty by design)
                    continue
                if off_streams + off + nbytes > job.cap:
                    failed = True
                    break
                native.memcpy_async(region.ptr + off_streams + off, job.dev.data_ptr() + off_streams + off, nbytes, "d2h",
                                    streams[r % 2].cuda_stream)
                total = off_streams + off + nbytes
            # (the last part's event has fired: every kernel of the store is done)
            st = self._release_pack_words(job)
            with self._lock:
                if st == 0 and not failed:
                    native.memcpy_async(region.ptr, job.dev.data_ptr(), off_streams, "d2h", streams[0].cuda_stream)
                ev2 = torch.cuda.Event()
                ev2.record(self.copy_stream2)
                self.copy_stream.wait_event(ev2)
                evd = torch.cuda.Event()
                evd.record(self.copy_stream)
                job.d2h_issued = True
                if job.dev is self._pack_dev:
                    self._pack_dev_free = evd
            evd.synchronize()
        if st or failed:
            arena.shrink(region, 0)
            raise native.NativeError("CacheGen store (pack): " + (native.describe_status(st) if st else "the device left no pack"))
        h = native.pack_info(region.ptr, total)  # the pack checks out where it lies now
        if int(h.total_bytes) != total:
            raise native.NativeError("CacheGen store (pack): the parts do not add up to the pack")
        return HostPack(arena.shrink(region, total), job.nchunks, job.chunk_tokens)

    def _release_pack_words(self, job: PackJob) -> int:
        """The store's kernels are done: its status word (returned), its size words and part words go back to their pools."""
        with self._lock:
            st, job.status_idx = self._status.read_release(job.status_idx), -1
            if job.sizes is not None:
                self._size_pool.append(job.sizes)
                job.sizes = None
            if job.part_info is not None:
                self._part_info_pool.append(job.part_info)
                job.part_info = None
            if st:
                job.d2h_issued = True
        return st

    def load_pack(self, pack: HostPack, chunk_begin: int, nchunks: int, dst: native.KVLayout, dst_tok0: int,
                  layers_per_range) -> DecodeJob:
        """Chunks [chunk_begin, chunk_begin + nchunks) of a pack -> decoded KV through ONE C-ABI call (lmc_load_pack): the streams of a
        range of layers are one contiguous transfer, the range's decode follows it, an event per range
        (DecodeJob.layer_events) lets the model run layer 0 while the later ranges are still crossing PCIe."""
        step = layers_per_range if isinstance(layers_per_range, int) or not layers_per_range else int(list(layers_per_range)[0])
        step = max(1, min(dst.L, int(step or dst.L)))
        ranges = layer_ranges(dst.L, step)
        with self._lock, torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            events = [native.NativeEvent() for _ in ranges]
            handles = (ctypes.c_void_p * len(events))(*[e.handle for e in events])
            st = self._status.acquire()
            try:
                self.ctx.load_pack(pack.blob.ptr, pack.blob.nbytes, chunk_begin, nchunks, dst, dst_tok0, step,
                                   ctypes.cast(handles, ctypes.c_void_p).value, stream=cur.cuda_stream,
                                   status_ptr=self._status.ptr(st))
            except BaseException:
                self._abandon_status(st, cur)
                raise
            return DecodeJob(events[-1], st, [(l1, ev) for (_, l1), ev in zip(ranges, events)], pool=self._status)

    def _dec_slots(self, n: int, stride: int, cur) -> torch.Tensor:
        if self._dec_arena is None or self._dec_arena.numel() < n * stride:
            self._dec_arena = torch.empty(n * stride, dtype=torch.uint8, device=self.device)
            # the caching allocator may hand out a block whose previous owner still has kernels queued on the
            # current stream: the side-stream copies into it must not start before those
            self.copy_stream.wait_stream(cur)
        return self._dec_arena

    def decode(self, host_blobs: Sequence, dst: native.KVLayout, dst_tok0: int, chunk_tokens: int,
               batch_chunks: Optional[int] = None) -> Optional[DecodeJob]:
        """H2D the blobs on the side stream and decode them on the current stream straight into `dst`,
        pipelined in batches of `decode_batch_chunks` (copy of batch b+1 overlaps the decode of batch b).
        host_blobs: HostBlob (pinned), bytes-like (staged through pinned memory) or uint8 CUDA tensors (blobs
        that are already in HBM: copied device to device).
        Asynchronous: returns a DecodeJob; finish_decode(job) waits for it and raises on a corrupt blob."""
        n = len(host_blobs)
        if n == 0:
            return None
        def _is_dev(b):
            return isinstance(b, torch.Tensor) and b.is_cuda

        sizes = [hb.nbytes if isinstance(hb, HostBlob) else (hb.numel() if _is_dev(hb) else len(hb)) for hb in host_blobs]
        stride = native.r16(max(sizes))
        with self._lock:
            with torch.cuda.device(self.device):
                cur = torch.cuda.current_stream(self.device)
                arena = self._dec_slots(n, stride, cur)
                st = self._status.acquire()
                try:
                    if self._dec_free is not None:
                        self.copy_stream.wait_event(self._dec_free)  # previous decode has read the slots
                    cs = self.copy_stream.cuda_stream
                    staged_off = 0
                    pageable = [hb for hb in host_blobs if not isinstance(hb, HostBlob) and not _is_dev(hb)]
                    if pageable:
                        need = sum(native.r16(len(b)) for b in pageable)
                        if self._stage is None or self._stage.nbytes < need:
                            self.copy_stream.synchronize()
                            self._stage = native.PinnedBuffer(need)
                        else:
                            self.copy_stream.synchronize()  # staging buffer is reused: earlier H2D must be done
                    # H2D and decode are pipelined in batches: the copy stream runs ahead, the compute stream
                    # decodes batch b as soon as its blobs have landed (one event per batch)
                    B = batch_chunks or self.decode_batch_chunks
                    last = None
                    for b0 in range(0, n, B):
                        b1 = min(n, b0 + B)
                        for i in range(b0, b1):
                            hb = host_blobs[i]
                            if _is_dev(hb):
                                # a blob already in HBM (xgmi:// connector): a device copy on the decoding stream itself,
                                # ordered behind whatever produced the tensor
                                native.memcpy_async(arena.data_ptr() + i * stride, hb.data_ptr(), sizes[i], "d2d",
                                                    cur.cuda_stream)
                                continue
                            if isinstance(hb, HostBlob):
                                src_ptr = hb.ptr
                            else:
                                data = hb if isinstance(hb, bytes) else bytes(hb)  # never mutates the caller's buffer
                                ctypes.memmove(self._stage.ptr + staged_off, data, len(data))
                                src_ptr = self._stage.ptr + staged_off
                                staged_off += native.r16(len(data))
                            native.memcpy_async(arena.data_ptr() + i * stride, src_ptr, sizes[i], "h2d", cs)
                        ready = torch.cuda.Event()
                        ready.record(self.copy_stream)
                        cur.wait_event(ready)
                        self.ctx.decode_chunks(arena.data_ptr() + b0 * stride, stride, b1 - b0, dst,
                                               dst_tok0 + b0 * chunk_tokens, chunk_tokens, stream=cur.cuda_stream,
                                               status_ptr=self._status.ptr(st))
                        last = torch.cuda.Event()
                        last.record(cur)
                    self._dec_free = last
                    return DecodeJob(last, st, pool=self._status)
                except BaseException:
                    self._abandon_status(st, cur)
                    raise

    def finish_decode(self, job: Optional[DecodeJob], what: str = "CacheGen decode") -> None:
        """Wait for THIS decode (its event, not the device) and raise NativeError if a kernel flagged its blobs
        (bad header / directory / stream): the destination then holds garbage and must not be used."""
        if job is None:
            return
        job.done.synchronize()
        st, job.status_idx = self._status.read_release(job.status_idx), -1
        meta = getattr(job, "_meta", None)
        if meta is not None:
            job._meta_pool.append(meta)
            job._meta = None
        if st:
            raise native.NativeError(f"{what}: " + native.describe_status(st))

    def close(self):
        with self._lock:
            for b in self._size_pool + [self._stage]:
                if b is not None:
                    b.free()
            self._size_pool = []
            self._stage = None
            self._enc_arena = self._dec_arena = None


_codecs = {}
_codecs_lock = threading.Lock()


def get_codec(device: Optional[int] = None) -> CacheGenDeviceCodec:
    dev = torch.cuda.current_device() if device is None else int(device)
    with _codecs_lock:
        if dev not in _codecs:
            _codecs[dev] = CacheGenDeviceCodec(dev)
        return _codecs[dev]
