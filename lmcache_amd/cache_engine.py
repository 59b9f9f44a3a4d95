"""LMCacheEngine -- the drop-in boundary.

Same public surface as the reference's lmcache/cache_engine.py:
  LMCacheEngine(config, metadata)                                        :18-35
  .store(tokens, kv_tensors_raw, skip_existing=True, blocking=True)      :230-287
  .retrieve(tokens, mask=None) -> (KVCache | (), ret_mask)               :293-381
  .close()                                                               :383
  LMCacheEngineBuilder.get_or_create / get / destroy                     :387-436

What stays Python, bit-exact with the reference: token chunking (:68-84), the
SHA-256 prefix-hash chain (:58-66, :86-96), key construction (:37-44), the
skip-existing scan (:183-208), the prefix-only / suffix-mask retrieve rules
(:323-357, :379).

What moved into HIP: the three full-size copies of store()
(_tuple_kv_to_blob :98-118, _slice_kv_at :131-161) and the concat of
retrieve() (:362-368) are gone -- chunks are gathered from / scattered to the
caller's per-layer tensors by the kernels (put_kv_range / get_kv_range), or by
one lmc_copy_kv pass per chunk for backends that only speak chunk tensors.
"""
import hashlib
import time
from typing import Dict, Iterable, List, Optional, Tuple, Union

import torch

from lmcache_amd import native
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend import CreateStorageBackend
from lmcache_amd.utils import CacheEngineKey, KVCache, _lmcache_nvtx_annotate

logger = init_logger(__name__)


def _token_dim(fmt: str) -> int:
    if fmt == "vllm":
        return 0
    if fmt == "huggingface":
        return 1
    raise ValueError(f"Invalid format: {fmt}")


class LMCacheEngine:
    def __init__(self, config: LMCacheEngineConfig, metadata: LMCacheEngineMetadata):
        self.config = config
        self.metadata = metadata
        self.chunk_size = config.chunk_size
        self.save_decode_cache = config.save_decode_cache
        self.engine_ = CreateStorageBackend(config, metadata)
        logger.debug("Current storage backend type %s", type(self.engine_))

    # ------------------------------------------------------------------ index (pure Python)
    def _make_key(self, chunk_hash: str, fmt: str) -> CacheEngineKey:
        m = self.metadata
        return CacheEngineKey(fmt, m.model_name, m.world_size, m.worker_id, chunk_hash)

    def _num_tokens_in_kv(self, kv_tensors: Union[KVCache, torch.Tensor], fmt: str) -> int:
        return kv_tensors[0][0].shape[_token_dim(fmt)]

    def _get_init_hash(self) -> str:
        return ""

    def _hash(self, tokens: torch.Tensor, prefix_hash: str) -> str:
        # the token bytes are dtype dependent (int64 little-endian for the usual LongTensor)
        return hashlib.sha256(prefix_hash.encode("ascii") + tokens.cpu().numpy().tobytes()).hexdigest()

    def _chunk_tokens(self, tokens: torch.Tensor) -> Iterable[torch.Tensor]:
        for start in range(0, len(tokens), self.chunk_size):
            yield tokens[start:start + self.chunk_size]

    def _prefix_hash(self, token_chunks: Iterable[torch.Tensor], num_skip_chunk: Optional[int] = 0) -> List[str]:
        running = self._get_init_hash()
        out = []
        for chunk in token_chunks:
            running = self._hash(chunk, running)
            out.append(running)
        return out[num_skip_chunk:]

    def _prefix_hashes_of(self, tokens: torch.Tensor, num_skip_chunk: int = 0) -> List[str]:
        """_prefix_hash(_chunk_tokens(tokens), num_skip_chunk) with one device-to-host conversion for the whole
        token tensor and no per-chunk copies (the same digests: the hash input is the chunk's bytes either way) --
        this runs in front of every store / retrieve, 64 chunks for a 16 k context.

        The chain of the LAST call is kept (token bytes + digests): a serving engine looks the same prompt up, retrieves
        it and stores it, and the next turn of a conversation extends it -- the chunks whose bytes equal the kept ones
        (a memcmp of 2 KB per chunk, ~10x cheaper than its SHA-256) take their digests from there, and the chain
        resumes at the first chunk that differs.  Same digests by construction: digest k is a function of bytes
        [0, end of chunk k) alone."""
        buf = tokens.detach().cpu().contiguous().numpy().tobytes()  # (bytes: slices of it compare by memcmp)
        step = self.chunk_size * tokens.element_size()
        running = self._get_init_hash()
        out: List[str] = []
        kept = getattr(self, "_hash_chain", None)
        start = 0
        if kept is not None and kept[0] == step:
            kbuf, khashes = kept[1], kept[2]
            n = min(len(buf), len(kbuf))
            whole = n // step  # (a shorter last chunk hashes different bytes: only whole chunks are compared)
            if whole and buf[:whole * step] == kbuf[:whole * step]:
                same = whole
            else:
                same = 0
                while same < whole and buf[same * step:(same + 1) * step] == kbuf[same * step:(same + 1) * step]:
                    same += 1
            if same:
                out = list(khashes[:same])
                running = out[-1]
                start = same * step
        for pos in range(start, len(buf), step):
            h = hashlib.sha256(running.encode("ascii"))
            h.update(buf[pos:pos + step])
            running = h.hexdigest()
            out.append(running)
        self._hash_chain = (step, buf, list(out))
        return out[num_skip_chunk:]

    def _first_missing_chunk(self, chunk_hashes: List[str], fmt: str) -> Optional[int]:
        """Index of the first chunk the backend does not hold (None: all present) -- the
        skip-existing scan of _make_chunks_skip_existing (:192-202)."""
        for idx, h in enumerate(chunk_hashes):
            if not self.engine_.contains(self._make_key(h, fmt)):
                return idx
        return None

    @staticmethod
    def plan_retrieve(num_tokens: int, chunk_size: int, mask_false_prefix: int, hit_chunks: int):
        """Pure arithmetic of retrieve() (:323-329, :360-365, :379): given how many leading tokens the
        mask skips and how many consecutive chunks (after the skipped whole chunks) hit, return
        (num_skip_chunk, extra_token_len, retrieved_token_count)."""
        num_skip_chunk = mask_false_prefix // chunk_size
        extra = mask_false_prefix - num_skip_chunk * chunk_size
        if hit_chunks == 0:
            return num_skip_chunk, extra, 0
        covered_end = min((num_skip_chunk + hit_chunks) * chunk_size, num_tokens)
        return num_skip_chunk, extra, covered_end - mask_false_prefix

    # ------------------------------------------------------------------ data movement helpers
    def _as_cuda_kv(self, kv: KVCache) -> KVCache:
        if all(k.is_cuda and v.is_cuda for k, v in kv):
            return kv
        return tuple((k.cuda(), v.cuda()) for k, v in kv)

    def _blob_to_tuple_kv(self, blob: torch.Tensor) -> KVCache:
        # (one unbind per layer: layer[0], layer[1] are two indexing calls each -- 64 of them in front of a 32-layer model)
        return tuple(tuple(layer.unbind(0)) for layer in blob.unbind(0))

    # ------------------------------------------------------------------ store
    @_lmcache_nvtx_annotate
    @torch.no_grad()
    def store(self, tokens: torch.Tensor, kv_tensors_raw: KVCache, skip_existing=True, blocking=True) -> None:
        """tokens [seq_len]; kv_tensors_raw: per layer (K, V), [T,H,D] ("vllm") or [H,T,D] ("huggingface")."""
        fmt = self.metadata.fmt
        assert len(tokens.shape) == 1, f"Invalid shape of tokens: {tokens.shape}"
        assert len(kv_tensors_raw) > 0, "Empty kv_tensors"
        assert len(tokens) == self._num_tokens_in_kv(kv_tensors_raw, fmt), \
            "Number of tokens in the kv cache does not match the input tokens"
        self._store_from(tokens, lambda: native.KVLayout.from_kv_tuple(self._as_cuda_kv(kv_tensors_raw), fmt),
                         skip_existing, blocking)

    @_lmcache_nvtx_annotate
    @torch.no_grad()
    def store_paged(self, tokens: torch.Tensor, kv_caches, slot_mapping: torch.Tensor, block_size: int,
                    layout: str = "NBHD", skip_existing=True, blocking=True) -> None:
        """store() for a serving engine's PAGED KV cache -- what the external vLLM connector does around
        store() (`lmcache_store_kv`: gather the rows of every layer's cache by slot_mapping into [T,H,D], then
        store; docs/source/developer_tutorial/LLM_Engine.rst:91-122), without the gather copy: the kernels
        read the blocks where they lie.
          kv_caches     per layer a tensor [2, num_blocks, block_size, H, D] (layout "NBHD", vLLM's flash layout)
                        or [2, num_blocks, H, block_size, D] ("NHBD", BASELINE.json's north star), bf16 / fp16
          slot_mapping  int64 [len(tokens)]: token t lives in slot slot_mapping[t] = block * block_size + offset
        The engine's fmt must be "vllm" (chunks are keyed and laid out [L,2,T,H,D])."""
        assert self.metadata.fmt == "vllm", "paged KV is a vLLM layout"
        assert len(tokens.shape) == 1, f"Invalid shape of tokens: {tokens.shape}"
        assert len(kv_caches) > 0, "Empty kv_caches"
        assert len(tokens) == slot_mapping.numel(), "one slot per token"
        self._store_from(tokens, lambda: native.KVLayout.paged(kv_caches, slot_mapping, block_size, layout),
                         skip_existing, blocking)

    def _store_from(self, tokens: torch.Tensor, make_src, skip_existing: bool, blocking: bool) -> None:
        t_start = time.perf_counter()
        fmt = self.metadata.fmt
        ntok = len(tokens)
        cs = self.chunk_size
        chunk_hashes = self._prefix_hashes_of(tokens)
        first = 0
        if skip_existing:
            first = self._first_missing_chunk(chunk_hashes, fmt)
            if first is None:
                logger.info("Stored/updated 0 chunks (all present)")
                return
        keys = [self._make_key(h, fmt) for h in chunk_hashes[first:]]
        src = make_src()
        t_plan = time.perf_counter()
        if getattr(self.engine_, "supports_kv_layout", False):
            n = self.engine_.put_kv_range(keys, src, fmt, first * cs, ntok, cs, blocking=blocking)
        else:
            n = self.engine_.batched_put(self._gather_chunks(keys, src, fmt, first * cs, ntok), blocking=blocking)
        logger.info("Stored/updated %d chunks, total time %.4fs, planning %.4fs", n,
                    time.perf_counter() - t_start, t_plan - t_start)

    def _gather_chunks(self, keys, src: native.KVLayout, fmt: str, tok_begin: int, tok_end: int):
        """(key, contiguous chunk tensor) pairs for chunk-tensor backends: one lmc_copy_kv pass per chunk
        instead of stack + permute + split + contiguous (:98-118, :141-150)."""
        ctx = native.get_context(src.device.index)
        dt = native.torch_dtype(src.dtype)
        for i, key in enumerate(keys):
            t0 = tok_begin + i * self.chunk_size
            T = min(self.chunk_size, tok_end - t0)
            shape = (src.L, 2, T, src.H, src.D) if fmt == "vllm" else (src.L, 2, src.H, T, src.D)
            chunk = torch.empty(shape, dtype=dt, device=src.device)
            ctx.copy_kv(src, t0, T, native.KVLayout.from_chunk(chunk, fmt), 0)
            yield key, chunk

    # ------------------------------------------------------------------ retrieve
    @_lmcache_nvtx_annotate
    @torch.no_grad()
    def retrieve(self, tokens: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Tuple[KVCache, torch.Tensor]:
        """Prefix hits only; `mask` (suffix mask) marks the tokens whose KV is wanted.
        Returns (per-layer (K, V) tuple or (), ret_mask)."""
        fmt = self.metadata.fmt
        tdim = _token_dim(fmt)
        box = {}

        def make_dst(nret, L, H, D, dtype, dev):
            shape = (L, 2, nret, H, D) if fmt == "vllm" else (L, 2, H, nret, D)
            box["blob"] = torch.empty(shape, dtype=dtype, device=dev)
            return native.KVLayout.from_chunk(box["blob"], fmt)

        got, ret_mask = self._retrieve_into(tokens, mask, make_dst)
        if got == 0:
            return (), ret_mask
        blob = box["blob"].narrow(2 if fmt == "vllm" else 3, 0, got)
        ret = self._blob_to_tuple_kv(blob)
        assert ret[0][0].shape[tdim] == got
        return ret, ret_mask

    @_lmcache_nvtx_annotate
    @torch.no_grad()
    def retrieve_into_paged(self, tokens: torch.Tensor, kv_caches, slot_mapping: torch.Tensor, block_size: int,
                            layout: str = "NBHD", mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """retrieve() straight into a serving engine's PAGED KV cache: the decoded (or copied) KV of token t is
        written to slot slot_mapping[t] of every layer's cache -- the connector's `lmcache_retrieve_kv` +
        reshape_and_cache_flash scatter (LLM_Engine.rst:101-122) fused into the decode kernel's store.  Slots
        need not be contiguous or ordered (CacheBlend-style placement of a non-prefix segment, BASELINE
        configs[4]).  Returns ret_mask (True where KV was written); the caches of tokens outside it are untouched.
          slot_mapping  int64 [len(tokens)] (entries of tokens the mask skips are ignored)"""
        assert self.metadata.fmt == "vllm", "paged KV is a vLLM layout"
        assert len(tokens) == slot_mapping.numel(), "one slot per token"
        num_skip_tok = 0 if mask is None else int(len(mask) - int(torch.sum(mask)))

        def make_dst(nret, L, H, D, dtype, dev):
            c0 = kv_caches[0]
            assert len(kv_caches) == L and c0.dtype == dtype, "cache geometry / dtype differs from the stored chunks"
            return native.KVLayout.paged(kv_caches, slot_mapping[num_skip_tok:num_skip_tok + nret], block_size, layout)

        _, ret_mask = self._retrieve_into(tokens, mask, make_dst)
        return ret_mask

    @_lmcache_nvtx_annotate
    @torch.no_grad()
    def retrieve_layerwise(self, tokens: torch.Tensor, mask: Optional[torch.Tensor] = None,
                           layers_per_launch=1) -> "LayerwiseRetrieval":
        """retrieve() that does not make the model wait for the last layer: the KV tuple is returned at once and
        `layer_events` says when each range of layers is complete -- [(first layer after the range, event), ...] in
        layer order; the attention of layer l runs after `wait_layer(l)` (a stream-side wait).  With an HBM-resident
        CacheGen tier (local_device="cuda", local_serde="cachegen") the decode of layer l+1 then hides behind the
        model's layer l and a warm prefix costs the model almost nothing (bench.py: ttft_proxy).  Backends that
        cannot cut their retrieve by layer return one event that covers everything.  Call finish() before
        trusting the KV for good: it waits for the decode and raises if a stored blob was corrupt.
        layers_per_launch: a range size, or a schedule of range sizes whose last entry repeats, e.g. (2, 6, 24)."""
        fmt = self.metadata.fmt
        box, jobs = {}, []

        def make_dst(nret, L, H, D, dtype, dev):
            shape = (L, 2, nret, H, D) if fmt == "vllm" else (L, 2, H, nret, D)
            box["blob"] = torch.empty(shape, dtype=dtype, device=dev)
            box["L"] = L
            return native.KVLayout.from_chunk(box["blob"], fmt)

        got, ret_mask = self._retrieve_into(tokens, mask, make_dst, layers_per_launch=layers_per_launch, jobs_out=jobs)
        if got == 0:
            return LayerwiseRetrieval((), ret_mask, [], [])
        blob = box["blob"].narrow(2 if fmt == "vllm" else 3, 0, got)
        # one list of (first layer after the range, event) per decode job: a retrieve that spans several stores is
        # several runs (local_backend.get_kv_range), i.e. several jobs, and layer l is complete when EVERY job's range
        # that holds l is (round 4 kept only the last job's events and was right only because all jobs share a stream)
        event_sets = [list(job.layer_events) for _, job in jobs if job is not None and job.layer_events]
        if not event_sets or len(event_sets) < len(jobs):
            # some run of the retrieve (or all of it) carries no per-layer events -- a backend that finished or queued its
            # part in one piece: ONE event recorded now on the current stream, behind everything _retrieve_into queued,
            # covers those runs for every layer (wait_layer used to skip them, which was right only while
            # every run shared this stream AND stood in front of the other runs' events)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(blob.device))
            event_sets.append([(box["L"], ev)])
        # (the per-layer (K, V) views are made when `kv` is first read: 33 tensor calls that need not stand between the
        # launches above and the caller's first wait_layer)
        return LayerwiseRetrieval(lambda: self._blob_to_tuple_kv(blob), ret_mask, event_sets[-1], jobs, event_sets)

    def _retrieve_into(self, tokens: torch.Tensor, mask: Optional[torch.Tensor], make_dst,
                       layers_per_launch: Optional[int] = None, jobs_out: Optional[list] = None) -> Tuple[int, torch.Tensor]:
        """The body of retrieve(): prefix probe, the first-chunk trim of a suffix mask, then every hit chunk
        written to destination tokens 0 .. got-1 of the layout make_dst(nret, L, H, D, dtype, device) returns.
        -> (got = tokens written, ret_mask)."""
        t_start = time.perf_counter()
        fmt = self.metadata.fmt
        cs = self.chunk_size
        ret_mask = torch.ones_like(tokens, dtype=torch.bool)
        num_skip_tok = 0
        if mask is not None:
            num_skip_tok = int(len(mask) - int(torch.sum(mask)))
        num_skip_chunk = num_skip_tok // cs
        if num_skip_tok:
            ret_mask[:num_skip_tok] = False
        chunk_hashes = self._prefix_hashes_of(tokens, num_skip_chunk)
        # the key objects of the last few hash chains are kept (the chain's last digest names the whole chain): a warm
        # prefix is looked up, retrieved and stored with the same 64 keys (round 6: host time in front of the first launch)
        memo = getattr(self, "_key_memo", None)
        if memo is None:
            memo = self._key_memo = {}
        mk = (fmt, chunk_hashes[-1], len(chunk_hashes)) if chunk_hashes else None
        keys = memo.get(mk)
        if keys is None:
            keys = [self._make_key(h, fmt) for h in chunk_hashes]
            if mk is not None:
                if len(memo) >= 4:
                    memo.pop(next(iter(memo)))
                memo[mk] = keys
        dev = torch.device("cuda", torch.cuda.current_device())

        def miss():
            ret_mask[:] = False
            return 0, ret_mask

        if getattr(self.engine_, "supports_kv_layout", False):
            count = getattr(self.engine_, "contains_prefix", None)
            if count is not None:
                hits = count(keys)  # the same scan inside the backend: one call instead of one per key
            else:
                hits = 0
                for k in keys:
                    if not self.engine_.contains(k):
                        break
                    hits += 1
            _, extra, nret = self.plan_retrieve(len(tokens), cs, num_skip_tok, hits)
            if hits == 0 or nret <= 0:
                return miss()
            try:
                shape0, dtype = self.engine_.chunk_meta(keys[0])
            except KeyError:  # gone since `contains`: a miss
                return miss()
            L = shape0[0]
            H, D = (shape0[3], shape0[4]) if fmt == "vllm" else (shape0[2], shape0[4])
            dst = make_dst(nret, L, H, D, dtype, dev)
            try:
                if jobs_out is not None and getattr(self.engine_, "mode", None) in ("hbm-cachegen", "cachegen"):
                    got = self.engine_.get_kv_range(keys if hits == len(keys) else keys[:hits], dst, fmt, -extra, cs,
                                                    layers_per_launch=layers_per_launch, jobs_out=jobs_out)
                else:
                    got = self.engine_.get_kv_range(keys if hits == len(keys) else keys[:hits], dst, fmt, -extra, cs)
            except native.NativeError:
                # a stored blob that does not decode must never reach the model as KV: the whole lookup is a miss
                logger.exception("retrieve: a cached chunk failed to decode; treated as a miss")
                got = 0
            got = hits if got is None else got
            if got < hits:  # some chunks went missing after `contains`: keep the prefix that arrived
                hits = got
                _, extra, nret = self.plan_retrieve(len(tokens), cs, num_skip_tok, hits)
                if hits == 0 or nret <= 0:
                    return miss()
        else:
            chunks = []
            for chunk in self.engine_.batched_get(iter(keys)):
                if chunk is None:
                    break
                chunks.append(chunk)
            hits = len(chunks)
            _, extra, nret = self.plan_retrieve(len(tokens), cs, num_skip_tok, hits)
            if hits == 0 or nret <= 0:
                return miss()
            c0 = chunks[0]
            L = c0.shape[0]
            H, D = (c0.shape[3], c0.shape[4]) if fmt == "vllm" else (c0.shape[2], c0.shape[4])
            dst = make_dst(nret, L, H, D, c0.dtype, c0.device)
            ctx = native.get_context(c0.device.index)
            pos = -extra
            for c in chunks:  # scatter every chunk into its slice (replaces slice + torch.cat, :360-368)
                T = c.shape[2] if fmt == "vllm" else c.shape[3]
                skip = max(0, -pos)
                if skip < T:
                    ctx.copy_kv(native.KVLayout.from_chunk(c, fmt), skip, T - skip, dst, pos + skip)
                pos += T
        if num_skip_tok + nret < len(ret_mask):
            ret_mask[num_skip_tok + nret:] = False
        logger.info("Retrieved %d chunks (%d tokens in total) -- elapsed time %.4f", hits, nret,
                    time.perf_counter() - t_start)
        return nret, ret_mask

    def close(self):
        self.engine_.close()


class LayerwiseRetrieval:
    """What retrieve_layerwise returns: the KV tuple (being filled layer by layer), ret_mask, and the events."""

    def __init__(self, kv, ret_mask, layer_events, jobs, event_sets=None):
        # `layer_events` is the LAST run's list only (kept for callers of rounds 2-4): a retrieve that spans several
        # stores has one list per run, and layer l is complete when every run's range that holds l is -- use wait_layer()
        self._kv, self.ret_mask, self.layer_events, self._jobs = kv, ret_mask, layer_events, jobs
        self._event_sets = event_sets if event_sets is not None else ([layer_events] if layer_events else [])

    @property
    def kv(self):
        """The per-layer (K, V) tuple (being filled layer by layer); () on a miss."""
        if callable(self._kv):
            self._kv = self._kv()
        return self._kv

    def wait_layer(self, layer: int, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Make `stream` (default: the current one) wait until the KV of `layer` is complete.  No host wait."""
        if not self.layer_events:
            return
        st = stream or torch.cuda.current_stream()

        def wait(ev):  # a torch event, or a native.NativeEvent recorded by lmc_load_chunks
            if hasattr(ev, "handle"):
                ev.wait(st.cuda_stream)
            else:
                st.wait_event(ev)

        for events in self._event_sets:  # every decode job of the retrieve: the range of ITS events that holds `layer`
            for end, ev in events:
                if layer < end:
                    wait(ev)
                    break
            else:
                wait(events[-1][1])

    def finish(self) -> None:
        """Host-side completion: waits for the decode and raises NativeError if a blob did not check out."""
        jobs, self._jobs = self._jobs, []
        for codec, job in jobs:
            codec.finish_decode(job)


class LMCacheEngineBuilder:
    """Per-process registry of engines by instance id (cache_engine.py:387-436)."""
    _instances: Dict[str, LMCacheEngine] = {}
    _cfgs: Dict[str, LMCacheEngineConfig] = {}
    _metadatas: Dict[str, LMCacheEngineMetadata] = {}

    @classmethod
    def get_or_create(cls, instance_id: str, config: LMCacheEngineConfig,
                      metadata: LMCacheEngineMetadata) -> LMCacheEngine:
        if instance_id in cls._instances:
            if cls._cfgs[instance_id] != config or cls._metadatas[instance_id] != metadata:
                raise ValueError(f"Instance {instance_id} already exists with a different configuration or metadata.")
            return cls._instances[instance_id]
        engine = LMCacheEngine(config, metadata)
        cls._instances[instance_id] = engine
        cls._cfgs[instance_id] = config
        cls._metadatas[instance_id] = metadata
        return engine

    @classmethod
    def get(cls, instance_id: str) -> Optional[LMCacheEngine]:
        return cls._instances.get(instance_id)

    @classmethod
    def destroy(cls, instance_id: str) -> None:
        engine = cls._instances.pop(instance_id, None)
        if engine is not None:
            engine.close()
        cls._cfgs.pop(instance_id, None)
        cls._metadatas.pop(instance_id, None)
