"""LMCRemoteBackend -- serialise -> connector.set / connector.get -> deserialise.

Mirror of lmcache/storage_backend/remote_backend.py:23-180: the ONLY caller of
the CacheGen serde in the reference (put_blocking :119-126, get :154-168), with
the same async-put worker thread (:51-69) and existing-keys cache (:111-117).
The pipelined variant (:183-275) is LMCPipelinedRemoteBackend below.
"""
import queue
import threading
from typing import List, Optional, Set

import torch

from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.abstract_backend import LMCBackendInterface
from lmcache_amd.storage_backend.connector import CreateConnector
from lmcache_amd.storage_backend.serde import CreateSerde
from lmcache_amd.utils import CacheEngineKey, _lmcache_nvtx_annotate

logger = init_logger(__name__)


class RemoteBackendEndSignal:
    pass


class LMCRemoteBackend(LMCBackendInterface):
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        super().__init__()
        self.existing_keys: Set[CacheEngineKey] = set()
        self.put_thread = None
        assert config.remote_url is not None, "Need to provide remote_url when using LMCRemoteBackend"
        assert config.remote_serde is not None, "Need to provide remote_serde when using LMCRemoteBackend"
        self.connection = CreateConnector(config.remote_url)
        self.serializer, self.deserializer = CreateSerde(config.remote_serde, config, metadata)
        self.dst_device = "cuda"
        self._cuda_device = torch.cuda.current_device() if torch.cuda.is_available() else None
        self.put_queue: "queue.Queue" = queue.Queue()
        self.put_thread = threading.Thread(target=self.put_worker, daemon=True)
        self.put_thread.start()

    @_lmcache_nvtx_annotate
    def put_worker(self):
        if self._cuda_device is not None:
            torch.cuda.set_device(self._cuda_device)
        while True:
            item = self.put_queue.get()
            if isinstance(item, RemoteBackendEndSignal):
                break
            key, value = item
            try:
                self.put_blocking(key, value)
            except Exception:
                logger.exception("asynchronous remote put failed")

    def list(self) -> List[CacheEngineKey]:
        keys = [CacheEngineKey.from_string(k) for k in self.connection.list()]
        self.existing_keys.update(keys)
        return keys

    def contains(self, key: CacheEngineKey) -> bool:
        if key in self.existing_keys:
            return True
        if self.connection.exists(key.to_string()):
            self.existing_keys.add(key)
            return True
        return False

    def put_blocking(self, key: CacheEngineKey, kv_chunk: torch.Tensor) -> None:
        self.connection.set(key.to_string(), self.serializer.to_bytes(kv_chunk))
        self.existing_keys.add(key)

    def put(self, key: CacheEngineKey, kv_chunk: torch.Tensor, blocking: bool = True) -> None:
        if blocking:
            self.put_blocking(key, kv_chunk)
        else:
            self.put_queue.put((key, kv_chunk))

    @_lmcache_nvtx_annotate
    def get(self, key: CacheEngineKey) -> Optional[torch.Tensor]:
        if not self.contains(key):
            return None
        bs = self.connection.get(key.to_string())
        if bs is None or len(bs) == 0:
            return None
        return self.deserializer.from_bytes(bs).to(self.dst_device)

    def close(self):
        if self.put_thread is not None and self.put_thread.is_alive():
            self.put_queue.put(RemoteBackendEndSignal())
            self.put_thread.join()
        if self.connection is not None:
            self.connection.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LMCPipelinedRemoteBackend(LMCRemoteBackend):
    """Pipelined retrieve: mirror of LMCPipelinedRemoteBackend
    (lmcache/storage_backend/remote_backend.py:183-275).

    The reference runs a network thread and a deserialise thread joined through Python queues and
    returns `result_list` -- which silently drops misses, so results no longer line up with the keys
    (:224-226, :238-243).  Here:

      * one fetch thread pulls blobs from the connector (blocking I/O is what a thread is for);
      * everything after the bytes arrive is stream work, not thread work: blobs are staged through
        pinned memory, copied H2D on the side stream and decoded on the caller's stream, ordered by
        events (CacheGenDeviceCodec.decode) -- the fetch of blob k+1 overlaps the copy and the decode
        of blob k;
      * `batched_get` returns one entry PER KEY (None for a miss);
      * with remote_serde == "cachegen" the backend speaks the engine's range protocol
        (put_kv_range / chunk_meta / get_kv_range): chunks are encoded straight from the per-layer KV
        tensors and decoded straight into the caller's output tensor, no per-chunk tensors in between.
    """

    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        super().__init__(config, metadata)
        from concurrent.futures import ThreadPoolExecutor
        self.fmt = metadata.fmt
        self.supports_kv_layout = config.remote_serde == "cachegen"
        self.cachegen_config = None
        if self.supports_kv_layout:
            from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
            self.cachegen_config = CacheGenConfig.from_model_name(metadata.model_name)
        # a connector whose blobs live in device memory (xgmi://): chunks go encode arena -> owner's HBM -> decode
        # arena without a host hop
        self._dev_conn = hasattr(self.connection, "set_device") and self.supports_kv_layout
        self.fetch_batch = 8                      # blobs handed to one decode call at most
        self._fetcher = ThreadPoolExecutor(max_workers=1, thread_name_prefix="lmc-fetch")
        self._prefetched = {}                     # key -> bytes fetched by chunk_meta, consumed by the next read
        self._host_arena = None

    # ---- fetch stage ---------------------------------------------------------------------------------
    def _fetch_into(self, keys, out_q: "queue.Queue") -> None:
        try:
            for idx, key in enumerate(keys):
                bs = self._prefetched.pop(key, None)
                if bs is None and self.contains(key):
                    bs = self.connection.get(key.to_string())
                out_q.put((idx, bs if bs else None))
        except Exception as e:  # hand the failure to the consumer instead of dying silently
            out_q.put(e)
        out_q.put(RemoteBackendEndSignal())

    def _arrivals(self, keys):
        """Yield (idx, bytes-or-None, backlog) in key order while the fetch thread runs ahead; `backlog` tells
        whether more fetched items are already waiting (so a consumer can batch what has arrived)."""
        q: "queue.Queue" = queue.Queue()
        self._fetcher.submit(self._fetch_into, list(keys), q)
        while True:
            item = q.get()
            if isinstance(item, RemoteBackendEndSignal):
                return
            if isinstance(item, Exception):
                raise item
            yield item[0], item[1], not q.empty()

    # ---- reference API ---------------------------------------------------------------------------------
    @_lmcache_nvtx_annotate
    def batched_get(self, keys):
        results: List[Optional[torch.Tensor]] = []
        for _, bs, _ in self._arrivals(keys):
            results.append(None if bs is None else self.deserializer.from_bytes(bs).to(self.dst_device))
        return results

    # ---- range protocol (cachegen serde) ------------------------------------------------------------------
    def chunk_meta(self, key: CacheEngineKey):
        from lmcache_amd import native
        from lmcache_amd.storage_backend.serde.cachegen_decoder import output_spec
        if self._dev_conn:
            got = self.connection.peek(key.to_string(), native.HEADER_BYTES)
            if got is None:
                self.existing_keys.discard(key)
                raise KeyError(key)
            h = native.blob_info(got[0], total_len=got[1])
            return output_spec(self.fmt, h.num_layers, h.ntokens, h.num_heads, h.head_size)
        bs = self._prefetched.get(key)
        if bs is None:
            bs = self.connection.get(key.to_string())
            if not bs:
                self.existing_keys.discard(key)  # gone since `contains`: a miss, not an error
                raise KeyError(key)
            self._prefetched[key] = bs
        h = native.blob_info(bs)
        return output_spec(self.fmt, h.num_layers, h.ntokens, h.num_heads, h.head_size)

    def put_kv_range(self, keys, src, fmt: str, tok_begin: int, tok_end: int, chunk_tokens: int,
                     blocking: bool = True) -> int:
        from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
        n = (tok_end - tok_begin + chunk_tokens - 1) // chunk_tokens
        assert n == len(keys), "one key per chunk"
        if n == 0:
            return 0
        codec = get_codec(src.device.index)
        with torch.cuda.device(src.device):
            job = codec.encode(src, tok_begin, tok_end, chunk_tokens, self.cachegen_config.plane_bins(src.L))
            if self._dev_conn:  # device to device: every blob straight into its owner's HBM arena
                sizes = codec.sizes_of(job)
                for i, key in enumerate(keys):
                    self.connection.set_device(key.to_string(), job.arena[i * job.stride:i * job.stride + sizes[i]])
                    self.existing_keys.add(key)
                job.offload_issued = True  # the copies have completed (set_device publishes after they land)
                return n
        # drain the device arena now (the next encode reuses it); only the connector writes may be deferred
        if self._host_arena is None:
            self._host_arena = PinnedArena(slab_bytes=64 << 20)
        self._host_arena.reset()
        blobs, done = codec.offload(job, None, self._host_arena)  # range by range, overlapping the rest of the encode
        done.synchronize()
        payload = _DeferredSets([(k, hb.tobytes()) for k, hb in zip(keys, blobs)])
        if blocking:
            self._apply_sets(payload)
        else:
            self.put_queue.put(payload)
        return n

    def get_kv_range(self, keys, dst, fmt: str, dst_tok0: int, chunk_tokens: int) -> int:
        """Decode chunk i (stored under keys[i]) into dst tokens dst_tok0 + i*chunk_tokens ...  Returns the
        number of leading chunks written: a blob that has gone from the store since `contains` ends the run
        (the reference breaks on the first None chunk, cache_engine.py:339-345) and drops the key from the
        existing-keys cache.  Every fetched blob is checked on the host (header, geometry against `dst`, length)
        before it goes to the GPU; a blob whose streams do not check out there raises NativeError."""
        from lmcache_amd import native
        from lmcache_amd.storage_backend.serde.cachegen_device import get_codec
        codec = get_codec(dst.device.index)
        batch, first, jobs = [], 0, []
        keys = list(keys)

        def flush():
            nonlocal batch, first
            if batch:
                with torch.cuda.device(dst.device):
                    jobs.append(codec.decode(batch, dst, dst_tok0 + first * chunk_tokens, chunk_tokens))
                first += len(batch)
                batch = []

        def arrivals():
            if not self._dev_conn:
                yield from self._arrivals(keys)
                return
            for idx, key in enumerate(keys):  # blobs in device memory: no blocking I/O, so no fetch thread
                t = self.connection.get_device(key.to_string()) if self.contains(key) else None
                yield idx, t, idx + 1 < len(keys)

        try:
            for idx, bs, backlog in arrivals():
                if bs is None:
                    self.existing_keys.discard(keys[idx])
                    break
                if self._dev_conn:
                    h = native.blob_info(bs[:native.HEADER_BYTES].cpu().numpy().tobytes(), total_len=bs.numel())
                else:
                    h = native.blob_info(bs)  # raises NativeError on a bad header / truncated blob
                if (h.num_layers, h.num_heads, h.head_size) != (dst.L, dst.H, dst.D) or h.ntokens > chunk_tokens:
                    raise native.NativeError(f"chunk {idx}: blob geometry does not match the destination")
                batch.append(bs)
                # decode what has arrived as soon as the fetch thread falls behind, or a full batch is there
                if len(batch) >= self.fetch_batch or not backlog:
                    flush()
            flush()
        finally:
            err = None
            for j in jobs:  # wait for every decode that was launched, keep the first failure
                try:
                    codec.finish_decode(j)
                except native.NativeError as e:
                    err = err or e
            if err is not None:
                raise err
        return first

    def _apply_sets(self, item: "_DeferredSets") -> None:
        for key, bs in item.payload:
            self.connection.set(key.to_string(), bs)
            self.existing_keys.add(key)

    def put_worker(self):
        if self._cuda_device is not None:
            torch.cuda.set_device(self._cuda_device)
        while True:
            item = self.put_queue.get()
            if isinstance(item, RemoteBackendEndSignal):
                break
            try:
                if isinstance(item, _DeferredSets):
                    self._apply_sets(item)
                else:
                    key, value = item
                    self.put_blocking(key, value)
            except Exception:
                logger.exception("asynchronous remote put failed")

    def close(self):
        super().close()
        if getattr(self, "_fetcher", None) is not None:
            self._fetcher.shutdown(wait=True)
            self._fetcher = None
        if getattr(self, "_host_arena", None) is not None:
            self._host_arena.close()
            self._host_arena = None


class _DeferredSets:
    def __init__(self, payload):
        self.payload = payload
