/*
 * lmc_hip.h -- C ABI of liblmc_hip.so, the MI355X (gfx950) KV hot path.
 *
 * Drop-in boundary for LMCache's CacheGen serde + host-DRAM offload path.
 * Every entry point cites the reference interface it replaces (paths relative
 * to the reference tree).  Conventions:
 *   - plain C types only; pointers are DEVICE pointers unless suffixed _h
 *     (host) -- no torch types cross this boundary;
 *   - every call that touches the GPU takes the hipStream_t to run on
 *     (`lmc_stream_t`, NULL = default stream), is ASYNCHRONOUS and never calls
 *     hipDeviceSynchronize (the reference flags torch.cuda.synchronize() as
 *     harmful on this path, lmcache/storage_backend/local_backend.py:83-85);
 *   - return value: 0 = ok, negative = error (lmc_strerror); nothing throws;
 *   - re-entrant per context: calls on different streams may interleave; the
 *     context orders its internal workspace with events, not host syncs.
 *
 * Plane order everywhere: p = kv * L + layer (all K planes, then all V
 * planes) -- the order of the reference's encode_input / cdf tensors
 * (cachegen_encoder.py:284,290).
 */
#ifndef LMC_HIP_H
#define LMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "lmc_format.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmc_ctx lmc_ctx;
typedef void* lmc_stream_t; /* hipStream_t */
typedef void* lmc_event_t;  /* hipEvent_t  */

#define LMC_OK 0
#define LMC_ERR_INVALID (-1)     /* bad argument / unsupported geometry            */
#define LMC_ERR_HIP (-2)         /* a HIP runtime call failed (see lmc_last_hip_error) */
#define LMC_ERR_NOMEM (-3)
#define LMC_ERR_DEVICE_FLAG (-4) /* a kernel reported an error (lmc_device_status)  */

const char* lmc_strerror(int code);
/* hipError_t of the last failing HIP call made by this library on this thread. */
int lmc_last_hip_error(void);
/* ABI version of this header. */
int lmc_abi_version(void);
#define LMC_ABI_VERSION 6

/* ------------------------------------------------------------------ */
/* KV addressing                                                       */
/* ------------------------------------------------------------------ */
/*
 * Where the 16-bit element (layer l, kv, token t, head h, dim d) of a KV cache
 * lives.  One descriptor covers every layout the reference's engine and the
 * (external) vLLM connector hand over:
 *   - chunk blob  [L,2,T,H,D]  "vllm" fmt          (cache_engine.py:137-138)
 *   - chunk blob  [L,2,H,T,D]  "huggingface" fmt   (cache_engine.py:139-140)
 *   - the KVCache tuple of per-layer (K,V) tensors passed to
 *     LMCacheEngine.store (cache_engine.py:230-236) -> plane_ptrs
 *   - vLLM paged blocks addressed through slot_mapping
 *     (docs/source/developer_tutorial/LLM_Engine.rst:91-122)
 *
 *   addr = plane_base(l,kv) + tok_off(t) + h*stride_head + d        [elements]
 *   plane_base = plane_ptrs ? plane_ptrs[2*l+kv] : base + l*stride_layer + kv*stride_kv
 *   tok_off(t) = slot_mapping ? (s / block_size)*stride_block + (s % block_size)*stride_token,
 *                               s = slot_mapping[t]
 *                             : t*stride_token
 * d is contiguous.  A plane has a multiple of 8 channels (num_heads * head_size % 8 == 0).  Layouts the ENCODERS read
 * (lmc_quantize, lmc_encode_chunks, lmc_store_*) are read with 16-byte vectors of 8 channels: base 16-byte aligned,
 * stride_layer / stride_kv / stride_token / stride_block multiples of 8 elements, and either head_size % 8 == 0 with
 * stride_head % 8 == 0, or stride_head == head_size (the heads of a token row back to back -- the vllm chunk, the
 * per-layer [T,H,D] tensors, NBHD blocks -- where any head_size will do).  The DECODERS' destination and both sides of
 * lmc_copy_kv take any strides and any head_size (lmc_copy_kv copies element-wise when a side is not vector-readable:
 * the way to bring such a range into a chunk the encoders take; lmcache_amd's codec does that by itself).
 */
typedef struct lmc_kv_layout {
  int32_t dtype;      /* LMC_DTYPE_BF16 / LMC_DTYPE_FP16 */
  int32_t num_layers; /* L */
  int32_t num_heads;  /* H (KV heads held by this rank) */
  int32_t head_size;  /* D */
  const void* base;
  const void* const* plane_ptrs; /* device array [2L] of device pointers, or NULL */
  int64_t stride_layer;
  int64_t stride_kv;
  int64_t stride_token;
  int64_t stride_head;
  const int64_t* slot_mapping; /* device [ntokens] or NULL */
  int32_t block_size;
  int32_t _pad;
  int64_t stride_block;
} lmc_kv_layout;

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
/* Replaces the per-serializer device state of CacheGenSerializer /
 * CacheGenDeserializer.__init__ (cachegen_encoder.py:330-350,
 * cachegen_decoder.py:110-140: bins tensors, reusable output buffer). */
int lmc_ctx_create(int device, lmc_ctx** out);
int lmc_ctx_destroy(lmc_ctx* ctx);
/* Pre-size the encode workspace (symbols + padded stream scratch) for up to
 * max_chunks chunks of chunk_tokens tokens.  Optional: lmc_encode_chunks grows
 * it on demand (growth allocates, i.e. may stall that one call). */
int lmc_ctx_reserve(lmc_ctx* ctx, int L, int H, int D, int chunk_tokens, int max_chunks);
/* Status bits a kernel can raise (0 = ok).  A call that is given a `job_status` word reports into THAT
 * word (device-accessible uint32, e.g. pinned host memory, zeroed by the caller; valid once the call's
 * work has completed: after an event / stream sync by the caller), so that concurrent jobs never see each
 * other's failures; with job_status == NULL the bits go to the context's sticky word below. */
#define LMC_STATUS_STREAM_OVERFLOW 1u /* encode: a group stream outgrew its scratch slot (cannot happen) */
#define LMC_STATUS_BAD_HEADER 2u      /* decode: header / geometry / section offsets inconsistent */
#define LMC_STATUS_BAD_STREAM 4u      /* decode: directory out of bounds, words left over, final state wrong */
#define LMC_STATUS_LOOKBACK_TIMEOUT 8u
#define LMC_STATUS_BAD_SCALES 16u     /* decode: a plane's scales do not match their checksum (lmc_format.h: scsum) */
#define LMC_STATUS_HOST_ARENA_FULL 32u /* lmc_store_chunks: the pinned arena cannot take the job's blobs */
/* A decode that raises any bit leaves the destination rows of the blobs concerned UNDEFINED (some of a plane's
 * waves may have stored before another wave saw the damage: BAD_SCALES is found by the plane's first wave only,
 * BAD_STREAM at the end of a stream).  A caller that decodes into live storage -- the paged entry points -- must
 * treat those tokens as not retrieved; lmcache_amd's engine turns every non-zero status into a miss. */
/* The context's sticky status word.  `clear` resets it. */
int lmc_device_status(lmc_ctx* ctx, int clear);

/* Geometry limits of this build: 2L <= LMC_MAX_PLANES planes, C = H*D <= LMC_MAX_CHANNELS channels per
 * plane (4096 = a 32-head x 128 MHA model; every BASELINE.json config fits), C a multiple of 8,
 * chunks of 1 .. 65535 tokens.  Outside them the entry points return LMC_ERR_INVALID. */
#define LMC_MAX_PLANES 256
#define LMC_MAX_CHANNELS 4096
#define LMC_FUSED_MAX_CHANNELS 1024 /* widest plane the fused encode kernel takes (k_fused.h) */

/* Which kernels lmc_encode_chunks launches.  The fused kernel (one workgroup quantises, codes and places a
 * run of whole planes of a chunk: k_fused.h) covers the 256-token chunks of planes of up to
 * LMC_FUSED_MAX_CHANNELS channels; wider planes, other chunk lengths and a ragged last chunk take
 * k_quantize + k_cdf_encode whatever the setting.
 *   AUTO (default)  fused when the job has more (chunk, plane) pairs than the chip has workgroup slots
 *                   (4 per CU), where it is the faster of the two;
 *   TWO_KERNELS     never fused;     FUSED  fused for any job size.
 * The blobs are byte-identical either way; the switch exists for A/B timing and for the parity tests. */
#define LMC_ENCODE_PATH_AUTO 0
#define LMC_ENCODE_PATH_TWO_KERNELS 1
#define LMC_ENCODE_PATH_FUSED 2
int lmc_ctx_set_encode_path(lmc_ctx* ctx, int path);

/* Per-kernel timing of the NEXT lmc_encode_chunks / lmc_decode_chunks calls:
 * when enabled the call brackets each of its kernels with hipEvents on the
 * caller's stream.  lmc_ctx_profile_read (after the caller has synchronised
 * that stream) returns the durations in ms of the last profiled call, in
 * launch order (encode: k_encode_fused, or k_quantize, k_cdf_encode [which also compacts the streams
 * into the blob]; decode: k_decode) and the number of entries written (<= cap). */
int lmc_ctx_profile(lmc_ctx* ctx, int enable);
int lmc_ctx_profile_read(lmc_ctx* ctx, float* ms_out, int cap);

/* ------------------------------------------------------------------ */
/* encode side                                                         */
/* ------------------------------------------------------------------ */
/*
 * Quantise tokens [tok_begin, tok_begin+ntok) of `src`.
 * Replaces torch_quant_vectorized applied to K and V + the torch.cat that
 * builds encode_input (cachegen_encoder.py:40-61, 278-285).
 *   bins_h   host int32 [2L], plane order (key_bins ++ value_bins,
 *            cachegen_encoder.py:339-350)
 *   sym_out  int8  [2L][ntok][C]   (= encode_input)
 *   scale_out u16  [2L][ntok]      raw bits of src dtype (= max_tensors_key ++ max_tensors_value)
 */
int lmc_quantize(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t ntok,
                 const int32_t* bins_h, int8_t* sym_out, uint16_t* scale_out, lmc_stream_t stream);

/*
 * Replaces torchac_cuda.calculate_cdf(sym, max_bins) (call sites
 * cachegen_encoder.py:287-289): per-(plane,channel) histogram over tokens ->
 * 16-bit strictly increasing CDF.
 *   sym int8 [P][T][C], cdf_out u16 [P][C][max_bins+1]; max_bins must be 32.
 */
int lmc_calculate_cdf(lmc_ctx* ctx, const int8_t* sym, int32_t P, int32_t T, int32_t C, int32_t max_bins,
                      uint16_t* cdf_out, lmc_stream_t stream);

/*
 * Fused encode of consecutive token chunks: quantise + CDF + entropy-encode +
 * compaction + container, one blob per chunk.  Replaces, per chunk,
 * LMCacheEngine._slice_kv_at (cache_engine.py:131-161) +
 * CacheGenSerializer.to_bytes -> encode_function -> encode_ntokens/
 * torchac_cuda.encode_fast_new -> collect_bytes -> CacheGenGPUEncoderOutput
 * (cachegen_encoder.py:225-325, 352-389), minus the pickle/D2H which is
 * lmc_memcpy_async's job.
 *   chunk i covers tokens [tok_begin + i*chunk_tokens, min(+chunk_tokens, tok_end))
 *   blobs    device arena; blob i is written at blobs + i*blob_stride;
 *            blob_stride >= lmc_blob_bound(L, chunk_tokens, H, D), multiple of 16
 *   sizes    device-accessible uint32 [nchunks] (device or pinned host memory):
 *            receives total_bytes of each blob
 *   job_status  see LMC_STATUS_* above (may be NULL)
 */
int lmc_encode_chunks(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end,
                      int32_t chunk_tokens, const int32_t* bins_h, void* blobs, uint64_t blob_stride,
                      uint32_t* sizes, uint32_t* job_status, lmc_stream_t stream);

/* ------------------------------------------------------------------ */
/* decode side                                                         */
/* ------------------------------------------------------------------ */
/*
 * Fused decode of nchunks blobs straight into the destination layout:
 * entropy-decode + dequantise + cast + scatter.  Replaces, per chunk,
 * CacheGenDeserializer.from_bytes -> decode_function_gpu -> decode_chunk /
 * torchac_cuda.decode_fast_prefsum -> do_dequantize -> stack/reshape/permute/
 * .to(16-bit) (cachegen_decoder.py:24-35, 51-106, 142-202) and the engine's
 * torch.cat over chunks (cache_engine.py:362-368).
 *   blob i is read from blobs + i*blob_stride (device memory)
 *   token t of chunk i is written to dst token  dst_tok0 + i*chunk_tokens + t;
 *   tokens that land below 0 are dropped (the "drop extra tokens in the first
 *   chunk" rule of retrieve(), cache_engine.py:360-365: pass dst_tok0 = -skip)
 *   dst->dtype selects the output type (bf16 for "vllm", fp16 for
 *   "huggingface": cachegen_decoder.py:190-200)
 */
int lmc_decode_chunks(lmc_ctx* ctx, const void* blobs, uint64_t blob_stride, int32_t nchunks,
                      const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, uint32_t* job_status,
                      lmc_stream_t stream);

/*
 * The same decode for a RANGE OF LAYERS of blobs that lie anywhere in device memory: the launch handles the K and V
 * planes of layers [layer_begin, layer_begin + layer_count) of every chunk.  A retrieve cut into one launch per
 * layer range (an event after each) hands layer 0's KV to the model after 1/L of the decode instead of all of it
 * -- the streams of a plane are independent and the blob's stream directory ({beg, end} per stream) gives random access to them.  Stands
 * where the reference decodes and concatenates the whole context before the first layer can run
 * (cache_engine.py:339-381; the connector writes layer by layer afterwards, LLM_Engine.rst:101-122).
 *   blob_ptrs       device array [nchunks] of device pointers, blob i at blob_ptrs[i] (16-byte aligned)
 *   max_blob_bytes  upper bound of the blobs' sizes (a blob whose header claims more is rejected)
 */
int lmc_decode_chunks_layers(lmc_ctx* ctx, const void* const* blob_ptrs, uint64_t max_blob_bytes, int32_t nchunks,
                             const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t layer_begin,
                             int32_t layer_count, uint32_t* job_status, lmc_stream_t stream);

/*
 * ... and a whole SCHEDULE of such launches in one call (round 6): range i covers layers
 * [layer_ends[i - 1], layer_ends[i]) (layer_ends[-1] = 0, ascending, the last entry = num_layers), each launch is
 * followed by hipEventRecord(events[i]) on `stream`.  What engine.retrieve_layerwise() issued as one C call and one
 * event per range from Python (0.26 ms of host time in front of the first kernel of a warm 16 k prefix) -- the host
 * side of cache_engine.py:293-381 for the HBM-resident tier.  events may be NULL (no events), or hold NULL entries.
 */
int lmc_decode_chunks_schedule(lmc_ctx* ctx, const void* const* blob_ptrs, uint64_t max_blob_bytes, int32_t nchunks,
                               const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t nranges,
                               const int32_t* layer_ends_h, const lmc_event_t* events_h, uint32_t* job_status,
                               lmc_stream_t stream);

/* Entropy-decode only (debug / parity): blob -> sym_out int8 [P][T][C].
 * Stands where torchac_cuda.decode_fast_prefsum stands (cachegen_decoder.py:65-66). */
int lmc_decode_symbols(lmc_ctx* ctx, const void* blob, int32_t L, int32_t H, int32_t D, int8_t* sym_out,
                       lmc_stream_t stream);

/* ------------------------------------------------------------------ */
/* lossless gather / scatter (raw chunks)                              */
/* ------------------------------------------------------------------ */
/*
 * Copy tokens [tok_begin, +ntok) between an arbitrary KV layout and a
 * contiguous chunk.  Replaces _tuple_kv_to_blob + _slice_kv_at
 * (cache_engine.py:98-161) on the store side, torch.cat + _blob_to_tuple_kv
 * (cache_engine.py:120-129, 362-368) on the retrieve side, and the external
 * connector's slot_mapping gather / reshape_and_cache_flash scatter
 * (LLM_Engine.rst:91-122).  Both layouts share L, H, D and dtype.
 * dst token index = dst_tok0 + (t - tok_begin).
 */
int lmc_copy_kv(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t ntok,
                const lmc_kv_layout* dst, int32_t dst_tok0, lmc_stream_t stream);

/* ------------------------------------------------------------------ */
/* host DRAM offload plumbing                                          */
/* ------------------------------------------------------------------ */
/* Pinned, device-accessible host memory (hipHostMalloc).  Replaces the
 * pageable kv_chunk.to("cpu") of LMCLocalBackend.put_blocking and its stubbed
 * pin-memory path (local_backend.py:50,82-100). */
int lmc_pinned_alloc(size_t bytes, void** out_h);
int lmc_pinned_free(void* ptr_h);
/* hipMemcpyAsync on `stream`; kind: 0 = D2H, 1 = H2D, 2 = D2D. */
int lmc_memcpy_async(void* dst, const void* src, size_t bytes, int kind, lmc_stream_t stream);

/*
 * The store leg in ONE call: encode the chunks (as lmc_encode_chunks does, into an arena the context owns) and move
 * every blob, at its exact size, into a pinned host arena -- with no host wait anywhere: the sizes are read on the GPU
 * by the copy kernel (k_offload.h), which also writes them, and the blobs' offsets, to pinned words for later.
 * Stands where LMCLocalBackend.put_blocking / put_nonblocking stand (local_backend.py:82-100): `.to("cpu")` behind a
 * device synchronisation.  A long job leaves in a few parts: the copy of part k (on a stream of the context) runs
 * beside the encode of part k + 1; `stream` is done when everything has landed.
 *   host_arena_h   pinned, device-mapped (lmc_pinned_alloc), host_cap bytes
 *   offsets_h      pinned uint64 [nchunks + 1]: blob i lies at host_arena_h + offsets_h[i]; [nchunks] = bytes used
 *   sizes_h        pinned uint32 [nchunks]: its size (0 and LMC_STATUS_HOST_ARENA_FULL if it did not fit)
 *   job_status     pinned status word of this job (may be NULL: the context's sticky word)
 * All three arrays are valid once `stream` has completed.
 */
int lmc_store_chunks(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                     const int32_t* bins_h, void* host_arena_h, uint64_t host_cap, uint64_t* offsets_h,
                     uint32_t* sizes_h, uint32_t* job_status, lmc_stream_t stream);

/*
 * The retrieve leg in ONE call, cut by layers: blobs in pinned host memory -> decoded KV in `dst`.  The blobs' headers
 * are checked by the CPU where they lie (pinned host memory: no GPU work, no wait), every blob goes over PCIe as one
 * hipMemcpyAsync (two copy streams of the context), and behind the transfer the decode runs as one launch per range of
 * `layers_per_range` layers (lmc_decode_chunks_layers) on `stream`, each followed by its event: the caller's model
 * starts on the first layers while the later ranges still decode.  (Per-range transfers -- the K run and the V run of
 * every blob, per range -- were measured slower than whole blobs: many short copies, see lmc_api.hip.)  Stands where
 * LMCLocalBackend.get + CacheGenDeserializer.from_bytes + the engine's torch.cat stand (local_backend.py:128-144,
 * cache_engine.py:339-381): whole chunks to the GPU one `.to("cuda")` at a time, then everything decoded and
 * concatenated, then the first layer can run.
 *   host_blob_ptrs_h  host array [nchunks] of pinned blob addresses (16-byte aligned); read during the call only
 *   sizes_h           host array [nchunks] of their sizes; read during the call only
 *   range_events      NULL, or [ceil(L / layers_per_range)] events: event r is recorded on `stream` behind the decode
 *                     of range r (the KV of its layers is complete once it has fired)
 *   layers_per_range  0 = all layers in one range
 * Returns LMC_ERR_INVALID, with nothing queued at all, if a blob's header does not check out (the stream directory is
 * checked by the decoder: LMC_STATUS_BAD_STREAM in the job's status word).
 * The blobs themselves must stay where they are until `stream` has completed.
 */
int lmc_load_chunks(lmc_ctx* ctx, const void* const* host_blob_ptrs_h, const uint32_t* sizes_h, int32_t nchunks,
                    const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t layers_per_range,
                    lmc_event_t* range_events, uint32_t* job_status, lmc_stream_t stream);

/*
 * Packs: the plane-major form of the pinned host tier (lmc_format.h, "pack").  lmc_store_pack is lmc_store_chunks with
 * the job's blobs written TRANSPOSED into one pinned region -- static sections of every chunk, then the streams ordered
 * (plane, chunk): K planes of every layer, then V planes -- so that lmc_load_pack moves the streams of a range of layers
 * as TWO hipMemcpyAsync (their K planes, their V planes) and the
 * model's first layers run while the later ranges are still crossing PCIe.  That is the order the consumer needs and
 * the reference cannot produce: its chunks arrive whole, one `.to("cuda")` each (local_backend.py:128-144), and
 * nothing can be decoded before the last one (cache_engine.py:339-381).
 *
 * lmc_store_pack: encode [tok_begin, tok_end) of `src` and write the pack to pack_h (pinned, device-mapped, pack_cap
 *   bytes; lmc_pack_bound is the worst case, a Llama-3-8B context needs about a quarter of it) without any host wait.
 *   sizes_h: pinned uint32 [nchunks], the blob sizes (0 = that chunk's encode failed).  Valid once `stream` has completed:
 *   the pack header's total_bytes (0 and LMC_STATUS_HOST_ARENA_FULL if the pack did not fit or a chunk failed).
 *   WHERE pack_h points decides who pays: mapped pinned HOST memory makes the call complete by itself (no host wait at
 *   all), but its copy kernel then posts PCIe writes from 64 workgroups for ~10 ms per 16 k context, and a bandwidth-bound
 *   kernel running beside it (a decode step) was measured 4.3x slower for that time -- the form for an otherwise idle
 *   GPU.  DEVICE memory (pack_h and sizes_h both device-accessible) builds the pack in HBM in ~0.3 ms; the caller reads
 *   total_bytes from the 256-byte header and moves the pack with DMA copies, which leave concurrent kernels alone
 *   (1.03x): what lmcache_amd's engine does (CacheGenDeviceCodec.store_pack).
 * lmc_pack_info: check a pack's header and offset table (host only), return the header.
 * lmc_pack_extract: chunk `chunk` of a pack as the blob lmc_encode_chunks wrote, byte for byte (host only: the
 *   one-chunk path of the backend, and how the tests pin a pack to the oracle).
 * lmc_load_pack: chunks [chunk_begin, chunk_begin + nchunks) of the pack (nchunks 0 = all that follow chunk_begin) ->
 *   decoded KV in `dst`, chunk chunk_begin + i at tokens dst_tok0 + i * chunk_tokens.  Offset table and static slots go first, then per range of `layers_per_range` layers
 *   (0 = all in one) the streams -- two copies per range for the whole pack, one per (layer, K/V) for a run of its
 *   chunks -- each followed by the range's decode on `stream` and, if given, range_events[r].  The pack must stay where it
 *   is until `stream` has completed.  LMC_ERR_INVALID, with nothing queued, if the pack does not check out.
 */
int lmc_store_pack(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                   const int32_t* bins_h, void* pack_h, uint64_t pack_cap, uint32_t* sizes_h, uint32_t* job_status,
                   lmc_stream_t stream);
/*
 * lmc_store_pack with the encode launched in `nparts` ranges of planes, each range packed as soon as it is coded
 * (round 6): after part r the streams region holds part r's segments at their final offsets, part_info_h[2 r] =
 * the part's offset in the streams region, part_info_h[2 r + 1] = its bytes (0: the pack has failed), and
 * part_events[r] is recorded on `stream` -- the caller moves [off_streams + offset, + bytes) to host memory with a
 * DMA copy while the later planes are still being encoded (CacheGenDeviceCodec.store_pack / finish_pack: a 16 k
 * store completes when its last PCIe byte lands, where lmc_store_pack + one copy of the finished pack waits for the
 * whole encode first).  Header, offset table and static slots ([0, off_streams)) are final after the LAST part.
 * pack_d must be DEVICE memory (a kernel that posts PCIe writes between the encode's parts would stall them);
 * part_info_h: pinned uint64 [2 nparts]; part_events: [nparts] events or NULL.  A job that cannot be split (a ragged
 * last chunk, a job too small for the fused encode) runs as ONE part: part_info_h[1 .. ) read {0, 0} and the
 * events of the unused parts are recorded behind the only one.  nparts <= 16.
 */
int lmc_store_pack_parts(lmc_ctx* ctx, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                         const int32_t* bins_h, void* pack_d, uint64_t pack_cap, uint32_t* sizes_h, int32_t nparts,
                         uint64_t* part_info_h, const lmc_event_t* part_events, uint32_t* job_status, lmc_stream_t stream);
int lmc_pack_info(const void* pack_h, uint64_t nbytes, lmc_pack_header* out);
int lmc_pack_extract(const void* pack_h, uint64_t nbytes, int32_t chunk, void* blob_out, uint64_t cap, uint32_t* size_out);
int lmc_load_pack(lmc_ctx* ctx, const void* pack_h, uint64_t pack_bytes, int32_t chunk_begin, int32_t nchunks,
                  const lmc_kv_layout* dst, int32_t dst_tok0, int32_t layers_per_range, lmc_event_t* range_events,
                  uint32_t* job_status, lmc_stream_t stream);


int lmc_stream_create(lmc_stream_t* out);
int lmc_stream_destroy(lmc_stream_t s);
int lmc_stream_synchronize(lmc_stream_t s);
int lmc_stream_wait_event(lmc_stream_t s, lmc_event_t e);
int lmc_event_create(lmc_event_t* out, int timing);
int lmc_event_destroy(lmc_event_t e);
int lmc_event_record(lmc_event_t e, lmc_stream_t s);
int lmc_event_synchronize(lmc_event_t e);
/* 1 = complete, 0 = pending, negative = error */
int lmc_event_query(lmc_event_t e);
int lmc_event_elapsed_ms(lmc_event_t start, lmc_event_t stop, float* ms);

/* Host-side blob header check/parse (no GPU).  Stands where
 * CacheGenEncoderOutput.from_bytes is used to inspect a blob
 * (tests/test_serde.py:60-62). */
int lmc_blob_info(const void* blob_h, size_t nbytes, lmc_blob_header* out);

#ifdef __cplusplus
}
#endif
#endif /* LMC_HIP_H */
