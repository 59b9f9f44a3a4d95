// lmc_api.hip -- C ABI (include/lmc_hip.h) over the gfx950 kernels.
//
// Host side only does argument checking, workspace management and kernel
// launches.  Nothing here synchronises the device; the workspace shared by
// successive encode calls is ordered with an event recorded after the last
// kernel of a job and waited on by the first kernel of the next one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>

#include "../../include/lmc_hip.h"
#include "k_copy.h"
#include "k_decode.h"
#include "k_encode_counts.h"
#include "k_fused.h"
#include "k_offload.h"
#include "k_quantize.h"

static thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                      \
  do {                                     \
    hipError_t e__ = (expr);               \
    if (e__ != hipSuccess) {               \
      g_last_hip = (int)e__;               \
      return LMC_ERR_HIP;                  \
    }                                      \
  } while (0)

struct lmc_ctx {
  int device;
  std::mutex mu;
  // Encode workspaces.  An encode job owns a workspace from its first kernel to its last; the job that takes it next
  // waits (on its own stream) for the event behind that last kernel.  TWO of them, so that a worker thread's
  // non-blocking put and the engine thread's put (the reference's model: local_backend.py:41-45, 72-80) run side by side
  // on their two streams instead of one behind the other: a job takes the slot its stream used last, else an idle one
  // (the second is allocated when the first is found busy by another stream for the first time), else the first.
  struct Workspace {
    u32* sym4 = nullptr;  size_t sym4_bytes = 0;       // symbols between the quantise stage and the coder
    u8* scratch = nullptr; size_t scratch_bytes = 0;   // two-kernel path: the streams before they are placed
    unsigned long long* agg = nullptr; size_t agg_bytes = 0;  // look-back granules of the in-kernel compaction
    hipEvent_t ws_free = nullptr;  // recorded after the last kernel that touches the workspace
    bool ws_used = false;
    hipStream_t last_stream = nullptr;
    u32* ticket = nullptr;     // device work-ticket counter of the coder launches (k_encode.h: EncodeArgs::ticket) ...
    u32 tickets_drawn = 0;     // ... and how many tickets the launches so far have drawn (mod 2^32): per workspace, since
                               // launches that share a counter must be ordered
  };
  Workspace ws[2];
  int num_cus = 256;
  int enc_path = LMC_ENCODE_PATH_AUTO;  // lmc_ctx_set_encode_path
  int fused_pl = 0;                     // LMC_FUSED_PL at lmc_ctx_create (A/B switch of tools/probes: planes per work item of narrow planes; 0 = the built-in choice)
  u32 epoch = 0;                        // of the last fused launch (tags its look-back granules)
  unsigned long long* pack_table = nullptr;  // device copy of a pack's offset table while it is written (lmc_store_pack)
  size_t pack_table_bytes = 0;
  // store / load legs (lmc_store_chunks, lmc_load_chunks)
  u8* store_arena = nullptr; size_t store_bytes = 0;  // blobs of the job being offloaded
  u8* load_slots = nullptr; size_t load_bytes = 0;    // HBM slots the gather kernel fills
  hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;  // the PCIe side of both legs (two DMA queues for the load)
  hipEvent_t store_free = nullptr, load_free = nullptr;  // the arena / the slots may be reused behind these
  bool store_used = false, load_used = false;
  hipEvent_t evpool[64] = {}; int evnext = 0;         // fork / join events of the two legs (round robin)
  u32* status_h = nullptr;  // pinned, device-accessible
  // optional per-kernel timing (lmc_ctx_profile)
  bool profile = false;
  hipEvent_t pev[8] = {};
  int pn = 0;  // events recorded by the last profiled call
};

// Record the next timing event of a profiled call (no-op when profiling is off).
static int prof_mark(lmc_ctx* c, hipStream_t s) {
  if (!c->profile) return LMC_OK;
  if (c->pn >= 8) return LMC_OK;
  if (!c->pev[c->pn]) HIP_TRY(hipEventCreate(&c->pev[c->pn]));
  HIP_TRY(hipEventRecord(c->pev[c->pn], s));
  c->pn++;
  return LMC_OK;
}

extern "C" {
#ifdef LMC_EXP_TIMELINE  // experiments only (not in include/lmc_hip.h): the fused kernel's phase time stamps
int lmc_debug_fused_timeline(void* host_out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_timeline), bytes < sizeof(g_fused_timeline) ? bytes : sizeof(g_fused_timeline));
}
int lmc_debug_decode_timeline(void* host_out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_decode_timeline), bytes < sizeof(g_decode_timeline) ? bytes : sizeof(g_decode_timeline));
}
#endif

const char* lmc_strerror(int code) {
  switch (code) {
    case LMC_OK: return "ok";
    case LMC_ERR_INVALID: return "invalid argument or unsupported geometry";
    case LMC_ERR_HIP: return "HIP runtime error (see lmc_last_hip_error)";
    case LMC_ERR_NOMEM: return "out of memory";
    case LMC_ERR_DEVICE_FLAG: return "a kernel flagged an error (see lmc_device_status)";
    default: return "unknown error";
  }
}
int lmc_last_hip_error(void) { return g_last_hip; }
int lmc_abi_version(void) { return LMC_ABI_VERSION; }

int lmc_ctx_create(int device, lmc_ctx** out) {
  if (!out) return LMC_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  lmc_ctx* c = new (std::nothrow) lmc_ctx();
  if (!c) return LMC_ERR_NOMEM;
  c->device = device;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->num_cus = v;
    if (const char* e = getenv("LMC_FUSED_PL")) c->fused_pl = atoi(e);  // read ONCE: never in the encode path
  }
  hipError_t e = hipHostMalloc((void**)&c->status_h, 64, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) { g_last_hip = (int)e; delete c; return LMC_ERR_HIP; }
  memset(c->status_h, 0, 64);
  for (int k = 0; k < 2; k++) {
    e = hipEventCreateWithFlags(&c->ws[k].ws_free, hipEventDisableTiming);
    if (e != hipSuccess) { g_last_hip = (int)e; (void)hipHostFree(c->status_h); delete c; return LMC_ERR_HIP; }
  }
  *out = c;
  return LMC_OK;
}

int lmc_ctx_destroy(lmc_ctx* c) {
  if (!c) return LMC_OK;
  (void)hipSetDevice(c->device);
  for (lmc_ctx::Workspace& w : c->ws) {
    if (w.ws_used) (void)hipEventSynchronize(w.ws_free);
    if (w.sym4) (void)hipFree(w.sym4);
    if (w.scratch) (void)hipFree(w.scratch);
    if (w.agg) (void)hipFree(w.agg);
    if (w.ticket) (void)hipFree(w.ticket);
    if (w.ws_free) (void)hipEventDestroy(w.ws_free);
  }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->copy_stream2) { (void)hipStreamSynchronize(c->copy_stream2); (void)hipStreamDestroy(c->copy_stream2); }
  if (c->store_arena) (void)hipFree(c->store_arena);
  if (c->load_slots) (void)hipFree(c->load_slots);
  if (c->pack_table) (void)hipFree(c->pack_table);
  if (c->store_free) (void)hipEventDestroy(c->store_free);
  if (c->load_free) (void)hipEventDestroy(c->load_free);
  for (int i = 0; i < 64; i++) if (c->evpool[i]) (void)hipEventDestroy(c->evpool[i]);
  for (int i = 0; i < 8; i++) if (c->pev[i]) (void)hipEventDestroy(c->pev[i]);
  if (c->status_h) (void)hipHostFree(c->status_h);
  delete c;
  return LMC_OK;
}

int lmc_ctx_profile(lmc_ctx* c, int enable) {
  if (!c) return LMC_ERR_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  c->profile = enable != 0;
  c->pn = 0;
  return LMC_OK;
}

int lmc_ctx_profile_read(lmc_ctx* c, float* ms_out, int cap) {
  if (!c || !ms_out || cap < 1) return LMC_ERR_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  int n = 0;
  for (int i = 0; i + 1 < c->pn && n < cap; i++, n++)
    HIP_TRY(hipEventElapsedTime(&ms_out[n], c->pev[i], c->pev[i + 1]));
  return n;
}

int lmc_ctx_set_encode_path(lmc_ctx* c, int path) {
  if (!c || path < LMC_ENCODE_PATH_AUTO || path > LMC_ENCODE_PATH_FUSED) return LMC_ERR_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  c->enc_path = path;
  return LMC_OK;
}

int lmc_device_status(lmc_ctx* c, int clear) {
  if (!c) return LMC_ERR_INVALID;
  int v = (int)__atomic_load_n(c->status_h, __ATOMIC_ACQUIRE);
  if (clear) __atomic_store_n(c->status_h, 0u, __ATOMIC_RELEASE);
  return v;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// `vec`: the layout is READ or WRITTEN with 16-byte vectors of 8 channels (the encoders, k_copy_kv): rows on 16-byte
// boundaries, and a vector must not leave its head -- head_size a multiple of 8, or the heads of a token row back to back
// (stride_head == head_size: the vllm chunk, the per-layer [T,H,D] tensors, NBHD paged blocks), where a vector that
// crosses a head boundary is still 8 consecutive elements.  !vec: element-wise access (the decoder's scatter, the
// element-wise copy): any strides.  Either way a plane has a multiple of 8 channels (the blob's geometry).
static bool layout_ok(const lmc_kv_layout* l, bool vec = true) {
  if (!l) return false;
  if (l->dtype != LMC_DTYPE_BF16 && l->dtype != LMC_DTYPE_FP16) return false;
  if (l->num_layers < 1 || 2 * l->num_layers > LMC_MAX_PLANES) return false;
  if (l->num_heads < 1 || l->head_size < 1) return false;
  const long long C = (long long)l->num_heads * l->head_size;
  if (C < 8 || (C & 7) || C > LMC_MAX_CHANNELS) return false;
  if (!l->base && !l->plane_ptrs) return false;
  if (l->slot_mapping && l->block_size < 1) return false;
  if (!vec) return true;
  if (l->head_size & 7) {
    if (l->stride_head != l->head_size) return false;
  } else if (l->stride_head & 7) return false;
  if (l->stride_token & 7) return false;
  if (!l->plane_ptrs && ((l->stride_layer & 7) || (l->stride_kv & 7) || ((uintptr_t)l->base & 15))) return false;
  if (l->slot_mapping && (l->stride_block & 7)) return false;
  return true;
}

static KvAddr to_addr(const lmc_kv_layout* l) {
  KvAddr a;
  a.base = (const u16*)l->base;
  a.plane_ptrs = (const u16* const*)l->plane_ptrs;
  a.slot_mapping = (const long long*)l->slot_mapping;
  a.stride_layer = l->stride_layer; a.stride_kv = l->stride_kv; a.stride_token = l->stride_token;
  a.stride_head = l->stride_head; a.stride_block = l->stride_block;
  a.block_size = l->block_size > 0 ? l->block_size : 1;
  a.L = l->num_layers; a.H = l->num_heads; a.D = l->head_size; a.dtype = l->dtype;
  return a;
}

static bool bins_ok(const int32_t* bins_h, int P, BinsArg* out) {
  if (!bins_h) return false;
  memset(out, 0, sizeof *out);
  for (int p = 0; p < P; p++) {
    // MAX = bins//2 - 1 >= 1 and symbols 0..2*MAX must fit the 32-entry CDF
    if (bins_h[p] < 4 || bins_h[p] > LMC_MAX_BINS) return false;
    out->b[p] = (u8)bins_h[p];
  }
  return true;
}

template <int DT, bool QUAD>
static int launch_quant_dt(const QuantArgs& a, hipStream_t s) {
  const int C = a.C;
  if (a.pc_limit < 1) return LMC_OK;
  const unsigned nz = (unsigned)((a.pc_limit + a.P - 1) / a.P);  // chunks that hold the plane-chunks to do
#define LQ(G, N)                                                                                   \
  do {                                                                                             \
    const int per_wg = 4 * (64 / (G)), TO = (a.TQ + 1) / 2; /* row octs per plane-chunk */         \
    dim3 grid((unsigned)((TO + per_wg - 1) / per_wg), (unsigned)a.P, nz);                        \
    hipLaunchKernelGGL((k_quantize<G, N, DT, QUAD>), grid, dim3(256), 0, s, a);                    \
  } while (0)
  // wide planes, workspace output: SPLIT waves share a row oct, 1024 channels each (k_quantize.h)
#define LQS(SPLIT)                                                                                 \
  do {                                                                                             \
    const int TO = (a.TQ + 1) / 2, per_wg = 4 / (SPLIT);                                           \
    dim3 grid((unsigned)((TO + per_wg - 1) / per_wg), (unsigned)a.P, nz);                        \
    hipLaunchKernelGGL((k_quantize<64, 2, DT, QUAD, QUAD ? SPLIT : 1>), grid, dim3(256), 0, s, a); \
  } while (0)
  if (C <= 128) LQ(16, 1);
  else if (C <= 256) LQ(32, 1);
  else if (C <= 512) LQ(64, 1);
  else if (C <= 1024) LQ(64, 2);
  else if (C <= 2048) { if (QUAD) LQS(2); else LQ(64, 4); }
  else if (C <= 4096) { if (QUAD) LQS(4); else LQ(64, 8); }
  else return LMC_ERR_INVALID;
#undef LQ
#undef LQS
  HIP_TRY(hipGetLastError());
  return LMC_OK;
}

template <bool QUAD>
static int launch_quant(const QuantArgs& a, hipStream_t s) {
  return a.src.dtype == LMC_DTYPE_BF16 ? launch_quant_dt<LMC_DTYPE_BF16, QUAD>(a, s)
                                       : launch_quant_dt<LMC_DTYPE_FP16, QUAD>(a, s);
}

static int ws_grow(void** p, size_t* have, size_t need) {
  if (*have >= need) return LMC_OK;
  if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) { g_last_hip = (int)e; return LMC_ERR_HIP; } *p = nullptr; *have = 0; }
  need = (need + (size_t)(1 << 20)) & ~(size_t)((1 << 20) - 1);
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess) { g_last_hip = (int)e; return e == hipErrorOutOfMemory ? LMC_ERR_NOMEM : LMC_ERR_HIP; }
  *have = need;
  return LMC_OK;
}

// The fused encode for a plane width (k_fused.h: lanes per quantise task, channel runs per lane, waves per row).
template <int DT, bool PSRC>
static void launch_fused_src(int C, dim3 grid, dim3 block, hipStream_t s, const FusedArgs& fa) {
  if (C <= 128) hipLaunchKernelGGL((k_encode_fused<16, 1, DT, FUSED_WAVES, PSRC>), grid, block, 0, s, fa);
  else if (C <= 256) hipLaunchKernelGGL((k_encode_fused<32, 1, DT, FUSED_WAVES, PSRC>), grid, block, 0, s, fa);
  else if (C <= 512) hipLaunchKernelGGL((k_encode_fused<64, 1, DT, FUSED_WAVES, PSRC>), grid, block, 0, s, fa);
  else hipLaunchKernelGGL((k_encode_fused<64, 2, DT, FUSED_WAVES, PSRC>), grid, block, 0, s, fa);  // C <= LMC_FUSED_MAX_CHANNELS
}
template <int DT>
static void launch_fused(int C, dim3 grid, dim3 block, hipStream_t s, const FusedArgs& fa) {
  // a paged source with a block size that is a power of two (every one vLLM offers) has kernel instances of its own: the
  // tokens' slots by scalar loads (k_fused.h, fused_tok_off)
  const u32 bs = (u32)fa.src.block_size;
  if (fa.src.slot_mapping && bs != 0u && (bs & (bs - 1u)) == 0u) launch_fused_src<DT, true>(C, grid, block, s, fa);
  else launch_fused_src<DT, false>(C, grid, block, s, fa);
}

// caller holds ctx->mu.  The symbol workspace and the look-back granules of `max_chunks` chunks, stream scratch for
// `scratch_chunks` of them (only the general coder launch -- chunk lengths other than 256, a ragged last chunk -- codes
// into scratch slots; the fused kernel and the counts-only coder launch code straight into the blobs).
static int reserve_locked(lmc_ctx::Workspace* c, int L, int H, int D, int chunk_tokens, int max_chunks, int scratch_chunks) {
  const size_t P = 2 * (size_t)L, C = (size_t)H * D, G = (C + 63) / 64, TQ = ((size_t)chunk_tokens + 3) / 4;
  const size_t need_sym = (size_t)max_chunks * P * TQ * C * 4;
  const size_t need_scr = (size_t)scratch_chunks * P * G * lmc_group_cap_bytes((uint32_t)chunk_tokens);
  const size_t need_agg = (size_t)max_chunks * P * G * 8;
  if (need_sym <= c->sym4_bytes && need_scr <= c->scratch_bytes && need_agg <= c->agg_bytes)
    return LMC_OK;
  // growing frees memory that queued kernels may still use: wait for them (this call only)
  if (c->ws_used) HIP_TRY(hipEventSynchronize(c->ws_free));
  int rc;
  if ((rc = ws_grow((void**)&c->sym4, &c->sym4_bytes, need_sym))) return rc;
  if ((rc = ws_grow((void**)&c->scratch, &c->scratch_bytes, need_scr))) return rc;
  const size_t agg_before = c->agg_bytes;
  if ((rc = ws_grow((void**)&c->agg, &c->agg_bytes, need_agg))) return rc;
  // fresh granules must not look like a published value of some epoch (k_fused.h)
  if (c->agg_bytes != agg_before) HIP_TRY(hipMemset(c->agg, 0, c->agg_bytes));
  return LMC_OK;
}

extern "C" {

int lmc_ctx_reserve(lmc_ctx* c, int L, int H, int D, int chunk_tokens, int max_chunks) {
  if (!c || L < 1 || H < 1 || D < 1 || (H * (long long)D) % 8 || chunk_tokens < 1 || max_chunks < 1) return LMC_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  std::lock_guard<std::mutex> lk(c->mu);
  return reserve_locked(&c->ws[0], L, H, D, chunk_tokens, max_chunks, max_chunks);
}

int lmc_quantize(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t ntok, const int32_t* bins_h,
                 int8_t* sym_out, uint16_t* scale_out, lmc_stream_t stream) {
  if (!c || !layout_ok(src) || ntok < 1 || ntok > 65535 || tok_begin < 0 || !sym_out || !scale_out) return LMC_ERR_INVALID;
  QuantArgs a;
  memset(&a, 0, sizeof a);
  a.src = to_addr(src);
  a.P = 2 * src->num_layers;
  if (!bins_ok(bins_h, a.P, &a.bins)) return LMC_ERR_INVALID;
  a.C = src->num_heads * src->head_size;
  a.tok_begin = tok_begin; a.tok_end = tok_begin + ntok; a.chunk_tokens = ntok; a.nchunks = 1;
  a.TQ = (ntok + 3) / 4;
  a.pc_limit = a.P;
  a.sym8 = sym_out;
  a.scale_base = (u8*)scale_out; a.scale_stride = 0;
  HIP_TRY(hipSetDevice(c->device));
  return launch_quant<false>(a, (hipStream_t)stream);
}

int lmc_calculate_cdf(lmc_ctx* c, const int8_t* sym, int32_t P, int32_t T, int32_t C, int32_t max_bins,
                      uint16_t* cdf_out, lmc_stream_t stream) {
  if (!c || !sym || !cdf_out || P < 1 || T < 1 || T > 65535 || C < 8 || (C & 7) || max_bins != LMC_MAX_BINS)
    return LMC_ERR_INVALID;
  EncodeArgs a;
  memset(&a, 0, sizeof a);
  a.sym8 = sym;
  a.tok_begin = 0; a.tok_end = T; a.chunk_tokens = T; a.nchunks = 1;
  a.P = P; a.C = C; a.G = (C + 63) / 64; a.TQ = (T + 3) / 4;
  a.cdf_out = cdf_out;
  a.status = c->status_h;
  HIP_TRY(hipSetDevice(c->device));
  long long n = (long long)P * a.G;
  hipLaunchKernelGGL((k_cdf_encode<false, false>), dim3((unsigned)((n + ENC_WAVES - 1) / ENC_WAVES)), dim3(64 * ENC_WAVES), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return LMC_OK;
}

// lmc_encode_chunks, optionally launched in `nparts` ranges of PLANES (all chunks of the job): after part r,
// after_part(r, planes coded so far, last part?) runs with the context lock held (lmc_store_pack_parts packs the range
// there).  Only a job that takes the fused kernel for every chunk is split; any other job is one part.
typedef std::function<int(int, int, bool)> AfterPart;
static int encode_chunks_parts(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                               const int32_t* bins_h, void* blobs, uint64_t blob_stride, uint32_t* sizes, uint32_t* job_status,
                               lmc_stream_t stream, int nparts, const AfterPart* after_part);

int lmc_encode_chunks(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                      const int32_t* bins_h, void* blobs, uint64_t blob_stride, uint32_t* sizes, uint32_t* job_status,
                      lmc_stream_t stream) {
  return encode_chunks_parts(c, src, tok_begin, tok_end, chunk_tokens, bins_h, blobs, blob_stride, sizes, job_status, stream, 1,
                             nullptr);
}

static int encode_chunks_parts(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                               const int32_t* bins_h, void* blobs, uint64_t blob_stride, uint32_t* sizes, uint32_t* job_status,
                               lmc_stream_t stream, int nparts, const AfterPart* after_part) {
  if (!c || !layout_ok(src) || tok_begin < 0 || tok_end <= tok_begin || chunk_tokens < 1 || chunk_tokens > 65535 ||
      !blobs || !sizes || ((uintptr_t)blobs & 15) || (blob_stride & 15))
    return LMC_ERR_INVALID;
  const int L = src->num_layers, H = src->num_heads, D = src->head_size;
  const int P = 2 * L, C = H * D, G = (C + 63) / 64;
  if (C > LMC_MAX_CHANNELS) return LMC_ERR_INVALID;
  const int nchunks = (tok_end - tok_begin + chunk_tokens - 1) / chunk_tokens;
  if (nchunks > 65535) return LMC_ERR_INVALID;  // chunks ride on gridDim.z of k_quantize
  if (blob_stride < lmc_blob_bound((uint32_t)L, (uint32_t)chunk_tokens, (uint32_t)H, (uint32_t)D)) return LMC_ERR_INVALID;
  BinsArg bins;
  if (!bins_ok(bins_h, P, &bins)) return LMC_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;

  std::lock_guard<std::mutex> lk(c->mu);
  bool split_done = false;  // the fused launch went out in plane ranges and after_part ran behind each
  // k_fused.h codes the 256-token chunks (the counts model) of every plane width: a work item is a run of whole planes
  // of one chunk -- 8 planes of <= 128 channels, 4 of <= 256, else one
  const int nfull = (tok_end - tok_begin) / chunk_tokens;  // chunks of exactly chunk_tokens tokens; a ragged one may follow
  // (planes of more than 1024 channels: two kernels -- the fused form built for them in round 4 never won below eight
  // generations of workgroups and was removed, k_fused.h)
  // chunk lengths: the counts model (and with it the counts-only coder launch) codes 2 .. 256 tokens, the fused kernel
  // takes the job's full chunks of 32 .. 256 tokens (round 5; both were 256 only)
  const bool fused_fits = chunk_tokens >= 32 && chunk_tokens <= (int)LMC_COUNTS_T && nfull > 0 && C <= LMC_FUSED_MAX_CHANNELS;
  // planes per work item: narrow planes are grouped until an item has about eight streams -- one per wave (C = 128,
  // 80 layers, 128 chunks, measured: 4 planes = 8 streams per item 0.65 ms, 8 planes 0.68, 2 planes 0.80)
  int pl = C <= 256 ? (8 / G > 1 ? 8 / G : 1) : 1;
  if (c->fused_pl >= 1 && c->fused_pl * G <= 16 && C <= 256) pl = c->fused_pl;  // A/B switch (tools/probes), read at lmc_ctx_create
  const int ipc = (P + pl - 1) / pl;
  // AUTO: the fused kernel pays once its workgroups fill the slots of the chip (4 per CU): measured on MI355X with
  // 64 planes of 1024 channels, 4 / 8 / 12 / 16 / 32 / 64 chunks: fused / two-kernel time = 1.27 / 1.16 / 1.05 / 1.00 /
  // 0.91 / 0.90 (tools/probes/encode_ab.hip)
  const long long auto_min = 4ll * c->num_cus;
  const bool fused = fused_fits && (c->enc_path == LMC_ENCODE_PATH_FUSED ||
                                    (c->enc_path == LMC_ENCODE_PATH_AUTO && (long long)nfull * ipc >= auto_min));
  // chunks that go through the general coder launch (scratch slots): every chunk of a job whose chunks are longer than
  // 256 tokens or whose streams do not fill 8-wave workgroups, a ragged last chunk of a single token
  const int tail_tokens = (tok_end - tok_begin) - nfull * chunk_tokens;  // 0: no ragged chunk
  const bool counts_geometry = chunk_tokens <= (int)LMC_COUNTS_T && ((long long)P * G) % 8 == 0;
  const bool general_only = !counts_geometry || chunk_tokens < (int)LMC_COUNTS_T_MIN;
  const bool tail_general = tail_tokens > 0 && (general_only || tail_tokens < (int)LMC_COUNTS_T_MIN);
  const int scratch_chunks = general_only ? (fused ? nchunks - nfull : nchunks) : (tail_general ? 1 : 0);
  // which workspace: the one this stream used last (stream order alone keeps the jobs apart), else an idle one, else --
  // both busy with other streams' jobs -- the first, behind its job
  lmc_ctx::Workspace* w = nullptr;
  for (lmc_ctx::Workspace& k : c->ws)
    if (!w && k.ws_used && k.last_stream == s) w = &k;
  for (lmc_ctx::Workspace& k : c->ws) {
    if (w) break;
    if (!k.ws_used) { w = &k; break; }
    const hipError_t q = hipEventQuery(k.ws_free);
    if (q == hipSuccess) w = &k;
    else (void)hipGetLastError();  // hipErrorNotReady
  }
  if (!w) w = &c->ws[0];
  int rc = reserve_locked(w, L, H, D, chunk_tokens, nchunks, scratch_chunks);
  if (rc) return rc;
  if (w->ws_used) HIP_TRY(hipStreamWaitEvent(s, w->ws_free, 0));
  // (A size word of 0 says "this chunk's encode did not finish" to whoever reads the words next: the kernels clear the
  // job's words themselves -- k_quantize on the two-kernel path, in the fused kernel the workgroup that later writes the
  // size -- so that no stale word of an earlier job stands in for a size and no memset dispatch sits in front of a job.)

  const int TQ = (chunk_tokens + 3) / 4;
  const u32 cap = lmc_group_cap_bytes((uint32_t)chunk_tokens);  // scratch slot stride (and capacity)
  lmc_blob_header hl;
  lmc_blob_layout((uint32_t)L, (uint32_t)chunk_tokens, (uint32_t)H, (uint32_t)D, &hl);
  const long long PG = (long long)P * G;

  EncodeArgs ea;
  memset(&ea, 0, sizeof ea);
  ea.sym4 = w->sym4;
  ea.tok_begin = tok_begin; ea.tok_end = tok_end; ea.chunk_tokens = chunk_tokens; ea.nchunks = nchunks;
  ea.P = P; ea.C = C; ea.G = G; ea.TQ = TQ;
  ea.sym_stride = (long long)TQ * C;
  ea.blobs = (u8*)blobs; ea.blob_stride = (long long)blob_stride;
  ea.scratch = w->scratch; ea.cap = cap;
  ea.status = job_status ? job_status : c->status_h;
  ea.bins = bins;
  ea.agg = w->agg; ea.sizes = sizes;
  ea.L = L; ea.H = H; ea.D = D; ea.dtype = src->dtype;
  u8* const scale_base = (u8*)blobs + hl.off_scales;  // off_scales does not depend on T

  if (!w->ticket) {
    HIP_TRY(hipMalloc((void**)&w->ticket, 64));
    HIP_TRY(hipMemsetAsync(w->ticket, 0, 64, s));  // ordered in front of the first launch that draws from it
    w->tickets_drawn = 0;
  }
  ea.ticket = w->ticket;
  c->pn = 0;
  // A launch that draws tickets and fails leaves the host's count and the device counter apart: both start over.
  auto tickets_reset = [&]() {
    (void)hipMemsetAsync(w->ticket, 0, 64, s);
    w->tickets_drawn = 0;
  };
  // the chunks [c0, c0 + n) of the job with the two kernels: k_quantize, then k_cdf_encode (CDF or counts table +
  // coder + in-kernel compaction of the streams into the blobs).  Measured alternatives that lost: HISTORY.md.
  auto two_kernels = [&](int c0, int n) -> int {
    EncodeArgs e2 = ea;
    e2.tok_begin = tok_begin + c0 * chunk_tokens; e2.nchunks = n;
    e2.sym4 = w->sym4 + (size_t)c0 * P * (size_t)ea.sym_stride;
    e2.blobs = (u8*)blobs + (size_t)c0 * blob_stride;
    e2.scratch = w->scratch;  // (at most one launch of a job codes into scratch: its slots are counted from its first chunk)
    e2.agg = w->agg + (size_t)c0 * PG;
    e2.sizes = sizes + c0;
    QuantArgs qa;
    memset(&qa, 0, sizeof qa);
    qa.src = to_addr(src); qa.bins = bins;
    qa.tok_begin = e2.tok_begin; qa.tok_end = tok_end; qa.chunk_tokens = chunk_tokens; qa.nchunks = n;
    qa.P = P; qa.C = C; qa.TQ = TQ; qa.pc_limit = n * P; qa.sym_stride = ea.sym_stride;
    qa.sym4 = const_cast<u32*>(e2.sym4);
    qa.scale_base = scale_base + (size_t)c0 * blob_stride;
    qa.scale_stride = (long long)blob_stride;
    qa.agg = e2.agg; qa.agg_n = (long long)n * PG;  // zeroed by the quantiser: one dispatch less per job
    qa.sizes = e2.sizes;
    int r;
    if ((r = prof_mark(c, s))) return r;
    if ((r = launch_quant<true>(qa, s))) return r;
    if ((r = prof_mark(c, s))) return r;
    const long long ngroups = (long long)n * PG;
    // launches whose chunks all have 2 .. 256 tokens (a ragged last chunk included) take the counts-only coder: 8 waves
    // per workgroup, 32 waves per CU
    const int last_tokens = std::min(chunk_tokens, tok_end - (e2.tok_begin + (n - 1) * chunk_tokens));
    const bool counts_only = !general_only && last_tokens >= (int)LMC_COUNTS_T_MIN;
    const int nw = counts_only ? 8 : ENC_WAVES;
    const unsigned nwg = (unsigned)((ngroups + nw - 1) / nw);
    e2.ticket_base = w->tickets_drawn;
    if (counts_only) hipLaunchKernelGGL((k_cdf_encode<true, true, 8, true>), dim3(nwg), dim3(64 * 8), 0, s, e2);
    else hipLaunchKernelGGL((k_cdf_encode<true, true>), dim3(nwg), dim3(64 * ENC_WAVES), 0, s, e2);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { g_last_hip = (int)le; tickets_reset(); return LMC_ERR_HIP; }
    w->tickets_drawn += nwg;  // only a launch that went out has drawn
    return prof_mark(c, s);
  };
  if (fused) {
    // One launch for the full chunks: a workgroup per (chunk, plane) quantises, then codes its streams into the blob.
    FusedArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.src = to_addr(src); fa.e = ea;
    fa.e.nchunks = nfull;
    fa.e.tok_end = tok_begin + nfull * chunk_tokens;
    fa.scale_base = scale_base; fa.scale_stride = (long long)blob_stride;
    c->epoch = (c->epoch + 1u) & 0x3fffffffu;
    if (!c->epoch) c->epoch = 1u;
    fa.epoch = c->epoch;
    fa.pl = pl; fa.ipc = ipc;
    // items are plane-major across the chunks (item = it * nfull + chunk): a range of `it` is a range of planes of ALL
    // chunks.  One launch, or -- a store that packs and ships the planes as they are coded -- `nparts` of them.
    const int np = (after_part && nfull == nchunks && nparts > 1) ? std::min(nparts, ipc) : 1;
    split_done = np > 1;
    const dim3 block(64 * FUSED_WAVES);
    if ((rc = prof_mark(c, s))) return rc;
    for (int r = 0; r < np; r++) {
      const int it0 = (int)((long long)ipc * r / np), it1 = (int)((long long)ipc * (r + 1) / np);
      const dim3 grid((unsigned)((long long)nfull * (it1 - it0)));
      fa.item_base = (u32)((long long)it0 * nfull);
      fa.e.ticket_base = w->tickets_drawn;
      if (src->dtype == LMC_DTYPE_BF16) launch_fused<LMC_DTYPE_BF16>(C, grid, block, s, fa);
      else launch_fused<LMC_DTYPE_FP16>(C, grid, block, s, fa);
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) { g_last_hip = (int)le; tickets_reset(); return LMC_ERR_HIP; }
      w->tickets_drawn += grid.x;
      if (np > 1 && (rc = (*after_part)(r, std::min(P, it1 * pl), r == np - 1))) return rc;
    }
    if ((rc = prof_mark(c, s))) return rc;
    if (nfull < nchunks && (rc = two_kernels(nfull, nchunks - nfull))) return rc;  // the ragged last chunk
  } else if (nfull > 0 && nfull < nchunks && !general_only && tail_general) {
    // the full chunks with the counts-only coder, a one-token last chunk with the general one
    if ((rc = two_kernels(0, nfull)) || (rc = two_kernels(nfull, nchunks - nfull))) return rc;
  } else {
    if ((rc = two_kernels(0, nchunks))) return rc;
  }

  if (after_part && !split_done && (rc = (*after_part)(0, P, true))) return rc;  // one part: everything is coded
  HIP_TRY(hipEventRecord(w->ws_free, s));
  w->ws_used = true;
  w->last_stream = s;
  return LMC_OK;
}

static int decode_common(lmc_ctx* c, const void* blobs, uint64_t blob_stride, int nchunks, int L, int H, int D,
                         uint32_t* job_status, DecodeArgs& a) {
  if (!c || !blobs || nchunks < 1 || ((uintptr_t)blobs & 15) || (blob_stride & 15)) return LMC_ERR_INVALID;
  a.blobs = (const u8*)blobs; a.blob_stride = (long long)blob_stride; a.nchunks = nchunks;
  a.blob_ptrs = nullptr; a.layer_begin = 0; a.layer_count = L;
  a.seg_off = nullptr; a.seg_streams = nullptr; a.seg_n = 0;
  a.P = 2 * L; a.C = H * D; a.G = (a.C + 63) / 64;
  // k_decode numbers its streams (chunk, plane, group) in 32 bits: 2^31 of them would be > 10^11 tokens in one call
  if ((long long)nchunks * a.P * a.G >= (1ll << 31)) return LMC_ERR_INVALID;
  a.status = job_status ? job_status : c->status_h;
  return LMC_OK;
}

static int decode_launch(lmc_ctx* c, DecodeArgs& a, const lmc_kv_layout* dst, hipStream_t hs) {
  int rc;
  const long long n = (long long)a.nchunks * 2 * a.layer_count * a.G;
  dim3 grid((unsigned)((n + DEC_WAVES - 1) / DEC_WAVES));
  std::lock_guard<std::mutex> lk(c->mu);
  c->pn = 0;
  if ((rc = prof_mark(c, hs))) return rc;
  const bool paged = dst->slot_mapping != nullptr;
  if (dst->dtype == LMC_DTYPE_BF16) {
    if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, true>), grid, dim3(64 * DEC_WAVES), 0, hs, a);
    else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, false>), grid, dim3(64 * DEC_WAVES), 0, hs, a);
  } else {
    if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, true>), grid, dim3(64 * DEC_WAVES), 0, hs, a);
    else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, false>), grid, dim3(64 * DEC_WAVES), 0, hs, a);
  }
  HIP_TRY(hipGetLastError());
  if ((rc = prof_mark(c, hs))) return rc;
  return LMC_OK;
}

int lmc_decode_chunks(lmc_ctx* c, const void* blobs, uint64_t blob_stride, int32_t nchunks, const lmc_kv_layout* dst,
                      int32_t dst_tok0, int32_t chunk_tokens, uint32_t* job_status, lmc_stream_t stream) {
  if (!layout_ok(dst, false) || chunk_tokens < 1) return LMC_ERR_INVALID;
  DecodeArgs a;
  memset(&a, 0, sizeof a);
  int rc = decode_common(c, blobs, blob_stride, nchunks, dst->num_layers, dst->num_heads, dst->head_size, job_status, a);
  if (rc) return rc;
  a.dst = to_addr(dst); a.dst_tok0 = dst_tok0; a.chunk_tokens = chunk_tokens;
  HIP_TRY(hipSetDevice(c->device));
  return decode_launch(c, a, dst, (hipStream_t)stream);
}

int lmc_decode_chunks_layers(lmc_ctx* c, const void* const* blob_ptrs, uint64_t max_blob_bytes, int32_t nchunks,
                             const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t layer_begin,
                             int32_t layer_count, uint32_t* job_status, lmc_stream_t stream) {
  if (!layout_ok(dst, false) || chunk_tokens < 1 || !blob_ptrs || layer_begin < 0 || layer_count < 1 ||
      layer_begin + layer_count > dst->num_layers)
    return LMC_ERR_INVALID;
  DecodeArgs a;
  memset(&a, 0, sizeof a);
  // `blobs` is only a non-null placeholder here: the kernel takes every blob's address from the table
  int rc = decode_common(c, blob_ptrs, (max_blob_bytes + 15) & ~(uint64_t)15, nchunks, dst->num_layers, dst->num_heads,
                         dst->head_size, job_status, a);
  if (rc) return rc;
  a.blob_ptrs = (const u8* const*)blob_ptrs;
  a.layer_begin = layer_begin; a.layer_count = layer_count;
  a.dst = to_addr(dst); a.dst_tok0 = dst_tok0; a.chunk_tokens = chunk_tokens;
  HIP_TRY(hipSetDevice(c->device));
  return decode_launch(c, a, dst, (hipStream_t)stream);
}

int lmc_decode_chunks_schedule(lmc_ctx* c, const void* const* blob_ptrs, uint64_t max_blob_bytes, int32_t nchunks,
                               const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t nranges,
                               const int32_t* layer_ends_h, const lmc_event_t* events_h, uint32_t* job_status,
                               lmc_stream_t stream) {
  if (!layout_ok(dst, false) || chunk_tokens < 1 || !blob_ptrs || nranges < 1 || !layer_ends_h) return LMC_ERR_INVALID;
  for (int i = 0, prev = 0; i < nranges; prev = layer_ends_h[i], i++)  // the whole schedule is checked before anything is launched
    if (layer_ends_h[i] <= prev || layer_ends_h[i] > dst->num_layers) return LMC_ERR_INVALID;
  if (layer_ends_h[nranges - 1] != dst->num_layers) return LMC_ERR_INVALID;
  DecodeArgs a;
  memset(&a, 0, sizeof a);
  int rc = decode_common(c, blob_ptrs, (max_blob_bytes + 15) & ~(uint64_t)15, nchunks, dst->num_layers, dst->num_heads,
                         dst->head_size, job_status, a);
  if (rc) return rc;
  a.blob_ptrs = (const u8* const*)blob_ptrs;
  a.dst = to_addr(dst); a.dst_tok0 = dst_tok0; a.chunk_tokens = chunk_tokens;
  HIP_TRY(hipSetDevice(c->device));
  for (int i = 0, prev = 0; i < nranges; prev = layer_ends_h[i], i++) {
    a.layer_begin = prev; a.layer_count = layer_ends_h[i] - prev;
    rc = decode_launch(c, a, dst, (hipStream_t)stream);
    if (rc) return rc;
    if (events_h && events_h[i]) HIP_TRY(hipEventRecord((hipEvent_t)events_h[i], (hipStream_t)stream));
  }
  return LMC_OK;
}

int lmc_decode_symbols(lmc_ctx* c, const void* blob, int32_t L, int32_t H, int32_t D, int8_t* sym_out,
                       lmc_stream_t stream) {
  if (!sym_out || L < 1 || H < 1 || D < 1 || (H * (long long)D) % 8) return LMC_ERR_INVALID;
  DecodeArgs a;
  memset(&a, 0, sizeof a);
  int rc = decode_common(c, blob, 0, 1, L, H, D, nullptr, a);
  if (rc) return rc;
  a.sym_out = sym_out;
  HIP_TRY(hipSetDevice(c->device));
  const long long n = (long long)a.P * a.G;  // every layer (decode_common set the full range)
  hipLaunchKernelGGL((k_decode<true, LMC_DTYPE_BF16, false>), dim3((unsigned)((n + DEC_WAVES - 1) / DEC_WAVES)), dim3(64 * DEC_WAVES), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return LMC_OK;
}

int lmc_copy_kv(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t ntok, const lmc_kv_layout* dst,
                int32_t dst_tok0, lmc_stream_t stream) {
  if (!c || !layout_ok(src, false) || !layout_ok(dst, false) || ntok < 1 || tok_begin < 0 || dst_tok0 < 0) return LMC_ERR_INVALID;
  if (src->num_layers != dst->num_layers || src->num_heads != dst->num_heads || src->head_size != dst->head_size ||
      src->dtype != dst->dtype)
    return LMC_ERR_INVALID;
  const bool vec = layout_ok(src) && layout_ok(dst);  // else one element per thread (any strides, any head_size)
  CopyArgs a;
  memset(&a, 0, sizeof a);
  a.src = to_addr(src); a.dst = to_addr(dst);
  a.tok_begin = tok_begin; a.ntok = ntok; a.dst_tok0 = dst_tok0;
  a.P = 2 * src->num_layers; a.C = src->num_heads * src->head_size;
  a.nvec = (long long)a.P * ntok * (vec ? a.C / 8 : a.C);
  HIP_TRY(hipSetDevice(c->device));
  long long blocks = (a.nvec + 255) / 256;
  if (blocks > 256LL * 64) blocks = 256LL * 64;  // grid-stride beyond 64 blocks per CU
  if (vec) hipLaunchKernelGGL(k_copy_kv, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_copy_kv_elem, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return LMC_OK;
}

// ---- host DRAM offload plumbing ---------------------------------------------
int lmc_pinned_alloc(size_t bytes, void** out_h) {
  if (!out_h || bytes == 0) return LMC_ERR_INVALID;
  hipError_t e = hipHostMalloc(out_h, bytes, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) { g_last_hip = (int)e; return e == hipErrorOutOfMemory ? LMC_ERR_NOMEM : LMC_ERR_HIP; }
  return LMC_OK;
}
int lmc_pinned_free(void* p) { if (p) HIP_TRY(hipHostFree(p)); return LMC_OK; }

int lmc_memcpy_async(void* dst, const void* src, size_t bytes, int kind, lmc_stream_t stream) {
  if (!dst || !src) return LMC_ERR_INVALID;
  if (bytes == 0) return LMC_OK;
  hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToHost : kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if (kind < 0 || kind > 2) return LMC_ERR_INVALID;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  return LMC_OK;
}

int lmc_stream_create(lmc_stream_t* out) {
  if (!out) return LMC_ERR_INVALID;
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = (lmc_stream_t)s;
  return LMC_OK;
}
int lmc_stream_destroy(lmc_stream_t s) { HIP_TRY(hipStreamDestroy((hipStream_t)s)); return LMC_OK; }
int lmc_stream_synchronize(lmc_stream_t s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return LMC_OK; }
int lmc_stream_wait_event(lmc_stream_t s, lmc_event_t e) { HIP_TRY(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); return LMC_OK; }
int lmc_event_create(lmc_event_t* out, int timing) {
  if (!out) return LMC_ERR_INVALID;
  hipEvent_t e;
  HIP_TRY(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
  *out = (lmc_event_t)e;
  return LMC_OK;
}
int lmc_event_destroy(lmc_event_t e) { HIP_TRY(hipEventDestroy((hipEvent_t)e)); return LMC_OK; }
int lmc_event_record(lmc_event_t e, lmc_stream_t s) { HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return LMC_OK; }
int lmc_event_synchronize(lmc_event_t e) { HIP_TRY(hipEventSynchronize((hipEvent_t)e)); return LMC_OK; }
int lmc_event_query(lmc_event_t e) {
  hipError_t r = hipEventQuery((hipEvent_t)e);
  if (r == hipSuccess) return 1;
  if (r == hipErrorNotReady) return 0;
  g_last_hip = (int)r;
  return LMC_ERR_HIP;
}
int lmc_event_elapsed_ms(lmc_event_t a, lmc_event_t b, float* ms) {
  if (!ms) return LMC_ERR_INVALID;
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return LMC_OK;
}

// ---- store / load legs ------------------------------------------------------------------------------------
int lmc_blob_info(const void* blob_h, size_t nbytes, lmc_blob_header* out);
static int legs_init(lmc_ctx* c) {  // caller holds c->mu
  if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!c->store_free) HIP_TRY(hipEventCreateWithFlags(&c->store_free, hipEventDisableTiming));
  if (!c->load_free) HIP_TRY(hipEventCreateWithFlags(&c->load_free, hipEventDisableTiming));
  return LMC_OK;
}
static int next_event(lmc_ctx* c, hipEvent_t* out) {  // caller holds c->mu; 64 events in rotation: a call uses <= 34
  hipEvent_t& e = c->evpool[c->evnext];
  c->evnext = (c->evnext + 1) % 64;
  if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out = e;
  return LMC_OK;
}

int lmc_store_chunks(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                     const int32_t* bins_h, void* host_arena_h, uint64_t host_cap, uint64_t* offsets_h,
                     uint32_t* sizes_h, uint32_t* job_status, lmc_stream_t stream) {
  if (!c || !layout_ok(src) || tok_begin < 0 || tok_end <= tok_begin || chunk_tokens < 1 || chunk_tokens > 65535 ||
      !host_arena_h || ((uintptr_t)host_arena_h & 15) || !offsets_h || !sizes_h)
    return LMC_ERR_INVALID;
  const int nchunks = (tok_end - tok_begin + chunk_tokens - 1) / chunk_tokens;
  if (nchunks > 65535) return LMC_ERR_INVALID;
  const uint64_t stride = (lmc_blob_bound((uint32_t)src->num_layers, (uint32_t)chunk_tokens, (uint32_t)src->num_heads,
                                          (uint32_t)src->head_size) + 15) & ~(uint64_t)15;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if ((rc = legs_init(c))) return rc;
    if (c->store_bytes < (size_t)nchunks * stride) {
      if (c->store_used) HIP_TRY(hipEventSynchronize(c->store_free));  // growing frees the arena: this call only
      if ((rc = ws_grow((void**)&c->store_arena, &c->store_bytes, (size_t)nchunks * stride))) return rc;
    }
    if (c->store_used) HIP_TRY(hipStreamWaitEvent(s, c->store_free, 0));  // the previous job's copies have read the arena
  }
  // a long job leaves in parts: the copy of part k runs on the context's copy stream beside the encode of part k + 1
  const int nparts = nchunks >= 16 ? 4 : 1, per = (nchunks + nparts - 1) / nparts;
  uint32_t* st = job_status ? job_status : c->status_h;
  for (int c0 = 0; c0 < nchunks; c0 += per) {
    const int c1 = c0 + per < nchunks ? c0 + per : nchunks;
    const int t1 = tok_begin + c1 * chunk_tokens < tok_end ? tok_begin + c1 * chunk_tokens : tok_end;
    if ((rc = lmc_encode_chunks(c, src, tok_begin + c0 * chunk_tokens, t1, chunk_tokens, bins_h, c->store_arena + (size_t)c0 * stride,
                                stride, sizes_h + c0, job_status, stream)))
      return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipEvent_t ev;
    if ((rc = next_event(c, &ev))) return rc;
    HIP_TRY(hipEventRecord(ev, s));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, ev, 0));
    OffloadArgs oa;
    memset(&oa, 0, sizeof oa);
    oa.blobs = c->store_arena; oa.stride = (long long)stride; oa.sizes_d = sizes_h; oa.nchunks = nchunks; oa.chunk0 = c0;
    oa.host = (u8*)host_arena_h; oa.cap = host_cap; oa.offsets_h = (unsigned long long*)offsets_h; oa.sizes_h = sizes_h;
    oa.status = st;
    hipLaunchKernelGGL(k_offload, dim3(8, (unsigned)(c1 - c0)), dim3(256), 0, c->copy_stream, oa);
    HIP_TRY(hipGetLastError());
  }
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipEventRecord(c->store_free, c->copy_stream));
  c->store_used = true;
  HIP_TRY(hipStreamWaitEvent(s, c->store_free, 0));  // the caller's stream is done when the blobs have landed
  return LMC_OK;
}

// Host-side view of a blob that lies in pinned host memory: checked header + section offsets (lmc_blob_info's
// checks, without the copy-out).
static bool host_blob_ok(const u8* b, uint32_t size, int L, int H, int D, uint32_t max_bytes, lmc_blob_header* h) {
  if (size < sizeof(lmc_blob_header) || ((uintptr_t)b & 15)) return false;
  memcpy(h, b, sizeof *h);
  if (lmc_blob_info(b, size, h) != LMC_OK) return false;
  return h->num_layers == (uint32_t)L && h->num_heads == (uint32_t)H && h->head_size == (uint32_t)D &&
         h->total_bytes == size && h->total_bytes <= max_bytes;
}

int lmc_load_chunks(lmc_ctx* c, const void* const* host_blob_ptrs_h, const uint32_t* sizes_h, int32_t nchunks,
                    const lmc_kv_layout* dst, int32_t dst_tok0, int32_t chunk_tokens, int32_t layers_per_range,
                    lmc_event_t* range_events, uint32_t* job_status, lmc_stream_t stream) {
  if (!c || !host_blob_ptrs_h || !sizes_h || nchunks < 1 || !layout_ok(dst, false) || chunk_tokens < 1 || layers_per_range < 0)
    return LMC_ERR_INVALID;
  const int L = dst->num_layers, H = dst->num_heads, D = dst->head_size;
  if (H * D > LMC_MAX_CHANNELS) return LMC_ERR_INVALID;
  const uint64_t stride = (lmc_blob_bound((uint32_t)L, (uint32_t)chunk_tokens, (uint32_t)H, (uint32_t)D) + 15) & ~(uint64_t)15;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const int step = layers_per_range > 0 && layers_per_range < L ? layers_per_range : L;
  std::lock_guard<std::mutex> lk(c->mu);
  if ((rc = legs_init(c))) return rc;
  if (!c->copy_stream2) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking));
  if (c->load_bytes < (size_t)nchunks * stride) {
    if (c->load_used) HIP_TRY(hipEventSynchronize(c->load_free));
    if ((rc = ws_grow((void**)&c->load_slots, &c->load_bytes, (size_t)nchunks * stride))) return rc;
  }
  hipStream_t cs[2] = {c->copy_stream, c->copy_stream2};  // two DMA queues (tools/probes/d2h_streams.py)
  if (c->load_used) {  // the previous load's decodes have read the slots
    HIP_TRY(hipStreamWaitEvent(cs[0], c->load_free, 0));
    HIP_TRY(hipStreamWaitEvent(cs[1], c->load_free, 0));
  }
  // The blobs lie in pinned HOST memory: their headers are checked right here, by the CPU (a blob that does not check
  // out fails the call before anything is queued), and every blob travels as ONE hipMemcpyAsync, alternating between
  // two DMA queues.  Layer-range-major copies (the K run and the V run of every chunk, per range) were built first and
  // measured: 576 copies of ~0.9 MB for four ranges cost 14.0 ms per 16 k context against 9.8 ms for these 64 whole
  // blobs (~5 us of submission per copy, and the DMA engines lose rate on short transfers) -- more than the overlap
  // of the first ranges' decode with the later ranges' transfer can give back (<= 0.75 ms: the decode of a 16 k
  // context is 1.0 ms in all).  So the transfer stays chunk-major and the ranges are separate decode launches behind
  // it: range r is ready 9.8 ms + (r + 1) x 1.0 ms / ranges after the call, the model can start on layer 0 while the
  // later ranges still decode, and the whole context is there after 10.8 ms.
  for (int i = 0; i < nchunks; i++) {
    lmc_blob_header h;
    if (!host_blob_ok((const u8*)host_blob_ptrs_h[i], sizes_h[i], L, H, D, (uint32_t)(stride > 0xffffffffull ? 0xffffffffu : stride), &h))
      return LMC_ERR_INVALID;
  }
  for (int i = 0; i < nchunks; i++)
    HIP_TRY(hipMemcpyAsync(c->load_slots + (size_t)i * stride, host_blob_ptrs_h[i], (sizes_h[i] + 15u) & ~15u,
                           hipMemcpyHostToDevice, cs[i & 1]));
  for (int q = 0; q < 2; q++) {  // the decodes wait for both queues
    hipEvent_t ev;
    if ((rc = next_event(c, &ev))) return rc;
    HIP_TRY(hipEventRecord(ev, cs[q]));
    HIP_TRY(hipStreamWaitEvent(s, ev, 0));
  }
  DecodeArgs da;
  memset(&da, 0, sizeof da);
  if ((rc = decode_common(c, c->load_slots, stride, nchunks, L, H, D, job_status, da))) return rc;
  da.dst = to_addr(dst); da.dst_tok0 = dst_tok0; da.chunk_tokens = chunk_tokens;
  int r = 0;
  for (int l0 = 0; l0 < L; l0 += step, r++) {
    const int n = l0 + step <= L ? step : L - l0;
    da.layer_begin = l0; da.layer_count = n;
    const long long nstreams = (long long)nchunks * 2 * n * da.G;
    const dim3 grid((unsigned)((nstreams + DEC_WAVES - 1) / DEC_WAVES));
    const bool paged = dst->slot_mapping != nullptr;
    if (dst->dtype == LMC_DTYPE_BF16) {
      if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, true>), grid, dim3(64 * DEC_WAVES), 0, s, da);
      else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, false>), grid, dim3(64 * DEC_WAVES), 0, s, da);
    } else {
      if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, true>), grid, dim3(64 * DEC_WAVES), 0, s, da);
      else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, false>), grid, dim3(64 * DEC_WAVES), 0, s, da);
    }
    HIP_TRY(hipGetLastError());
    if (range_events && range_events[r]) HIP_TRY(hipEventRecord((hipEvent_t)range_events[r], s));
  }
  HIP_TRY(hipEventRecord(c->load_free, s));
  c->load_used = true;
  return LMC_OK;
}

// ---- packs (lmc_format.h): the layer-major form of the pinned host tier ---------------------------------------------
// lmc_store_pack and lmc_store_pack_parts.  nparts == 0: the pack kernels run on the context's copy stream behind the
// whole encode (pack_h may be mapped host memory: their PCIe writes then stall nobody's encode); nparts >= 1: the
// encode goes out in plane ranges and each range is packed on `stream` right behind the launch that coded it.
static int store_pack_impl(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                           const int32_t* bins_h, void* pack_h, uint64_t pack_cap, uint32_t* sizes_h, int32_t nparts,
                           uint64_t* part_info_h, const lmc_event_t* part_events, uint32_t* job_status, lmc_stream_t stream) {
  if (!c || !layout_ok(src) || tok_begin < 0 || tok_end <= tok_begin || chunk_tokens < 1 || chunk_tokens > 65535 || !bins_h ||
      !pack_h || ((uintptr_t)pack_h & 15) || !sizes_h || nparts < 0 || nparts > 16 || (nparts > 0 && !part_info_h))
    return LMC_ERR_INVALID;
  const int nchunks = (tok_end - tok_begin + chunk_tokens - 1) / chunk_tokens;
  const int L = src->num_layers, P = 2 * L;
  if (nchunks > 65535 || (long long)P * nchunks > (1ll << 22)) return LMC_ERR_INVALID;
  for (int p = 0; p < P; p++)
    if (bins_h[p] < 4 || bins_h[p] > LMC_MAX_BINS) return LMC_ERR_INVALID;
  PackArgs pa;
  memset(&pa, 0, sizeof pa);
  lmc_pack_layout((uint32_t)nchunks, (uint32_t)L, (uint32_t)chunk_tokens, (uint32_t)src->num_heads, (uint32_t)src->head_size, &pa.hdr);
  pa.hdr.ntokens = (uint32_t)(tok_end - tok_begin);
  if (pa.hdr.off_streams > pack_cap) return LMC_ERR_INVALID;
  const uint64_t stride = (lmc_blob_bound((uint32_t)L, (uint32_t)chunk_tokens, (uint32_t)src->num_heads,
                                          (uint32_t)src->head_size) + 15) & ~(uint64_t)15;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const size_t table_bytes = 8 * ((size_t)P * nchunks + 2);  // the table, and the running total of the parts
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if ((rc = legs_init(c))) return rc;
    if (c->store_bytes < (size_t)nchunks * stride || c->pack_table_bytes < table_bytes) {
      if (c->store_used) HIP_TRY(hipEventSynchronize(c->store_free));  // growing frees the buffers: this call only
      if (c->store_bytes < (size_t)nchunks * stride &&
          (rc = ws_grow((void**)&c->store_arena, &c->store_bytes, (size_t)nchunks * stride)))
        return rc;
      if (c->pack_table_bytes < table_bytes && (rc = ws_grow((void**)&c->pack_table, &c->pack_table_bytes, table_bytes))) return rc;
    }
    if (c->store_used) HIP_TRY(hipStreamWaitEvent(s, c->store_free, 0));  // the previous job's copies have read the arena
  }
  pa.blobs = c->store_arena; pa.stride = (long long)stride; pa.sizes_d = sizes_h;
  pa.n = nchunks; pa.L = L; pa.G = (int)pa.hdr.ngroups;
  pa.host = (u8*)pack_h; pa.cap = pack_cap; pa.table_d = c->pack_table;
  pa.status = job_status ? job_status : c->status_h;
  if (nparts == 0) {
    // the whole job is encoded (1 ms per 16 k tokens) before the first byte leaves: the pack kernels on the copy stream
    if ((rc = lmc_encode_chunks(c, src, tok_begin, tok_end, chunk_tokens, bins_h, c->store_arena, stride, sizes_h, job_status, stream)))
      return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipEvent_t ev;
    if ((rc = next_event(c, &ev))) return rc;
    HIP_TRY(hipEventRecord(ev, s));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, ev, 0));
    pa.p_begin = 0; pa.p_end = P; pa.last = 1; pa.part_h = nullptr;
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(256), 0, c->copy_stream, pa);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_pack_copy, dim3(64), dim3(256), 0, c->copy_stream, pa);  // PCIe-bound: 64 workgroups fill the link
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->store_free, c->copy_stream));
    c->store_used = true;
    HIP_TRY(hipStreamWaitEvent(s, c->store_free, 0));  // the caller's stream is done when the pack has landed
    return LMC_OK;
  }
  // ---- in parts: plane range r is packed on `stream` behind the launch that coded it.  A plane's end is the `beg` its
  // successor's first stream writes, so a part that is not the last packs the planes coded so far BUT the newest one,
  // which rides with the next part (1 / 64 of the data at Llama-3-8B's 64 planes).
  for (int r = 0; r < nparts; r++) part_info_h[2 * r] = part_info_h[2 * r + 1] = 0;  // (pinned host words: plain stores)
  int packed = 0, parts_out = 0;
  const AfterPart after = [&](int r, int coded, bool last) -> int {
    (void)r;
    const int upto = last ? P : coded - 1;
    if (upto > packed || last) {
      PackArgs q = pa;
      q.p_begin = packed; q.p_end = upto; q.last = last ? 1 : 0;
      q.part_h = (unsigned long long*)(part_info_h + 2 * parts_out);
      hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(256), 0, s, q);
      HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(k_pack_copy, dim3(64), dim3(256), 0, s, q);
      HIP_TRY(hipGetLastError());
      packed = upto;
    }
    if (part_events && part_events[parts_out]) HIP_TRY(hipEventRecord((hipEvent_t)part_events[parts_out], s));
    parts_out++;
    return LMC_OK;
  };
  if ((rc = encode_chunks_parts(c, src, tok_begin, tok_end, chunk_tokens, bins_h, c->store_arena, stride, sizes_h, job_status,
                                stream, nparts, &after)))
    return rc;
  std::lock_guard<std::mutex> lk(c->mu);
  for (; parts_out < nparts; parts_out++)  // a job that could not be split: the unused parts' events fire behind the only one
    if (part_events && part_events[parts_out]) HIP_TRY(hipEventRecord((hipEvent_t)part_events[parts_out], s));
  HIP_TRY(hipEventRecord(c->store_free, s));
  c->store_used = true;
  return LMC_OK;
}

int lmc_store_pack(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                   const int32_t* bins_h, void* pack_h, uint64_t pack_cap, uint32_t* sizes_h, uint32_t* job_status,
                   lmc_stream_t stream) {
  return store_pack_impl(c, src, tok_begin, tok_end, chunk_tokens, bins_h, pack_h, pack_cap, sizes_h, 0, nullptr, nullptr,
                         job_status, stream);
}

int lmc_store_pack_parts(lmc_ctx* c, const lmc_kv_layout* src, int32_t tok_begin, int32_t tok_end, int32_t chunk_tokens,
                         const int32_t* bins_h, void* pack_d, uint64_t pack_cap, uint32_t* sizes_h, int32_t nparts,
                         uint64_t* part_info_h, const lmc_event_t* part_events, uint32_t* job_status, lmc_stream_t stream) {
  if (nparts < 1) return LMC_ERR_INVALID;
  return store_pack_impl(c, src, tok_begin, tok_end, chunk_tokens, bins_h, pack_d, pack_cap, sizes_h, nparts, part_info_h,
                         part_events, job_status, stream);
}

// Host-side check of a pack that lies in (pinned) host memory.
static bool pack_ok(const u8* b, uint64_t nbytes, lmc_pack_header* h) {
  if (!b || ((uintptr_t)b & 15) || nbytes < LMC_PACK_HEADER_BYTES) return false;
  memcpy(h, b, sizeof *h);
  if (h->magic != LMC_PACK_MAGIC || h->version != LMC_PACK_VERSION || h->header_bytes != LMC_PACK_HEADER_BYTES) return false;
  if (h->nchunks < 1 || h->nchunks > 65535 || h->num_layers < 1 || h->num_layers > LMC_MAX_PLANES / 2 || h->num_heads == 0 ||
      h->head_size == 0 || h->chunk_tokens < 1 || h->chunk_tokens > 65535 || (h->static_stride & 15) ||
      h->static_stride < sizeof(lmc_blob_header))
    return false;
  const uint64_t N = 2ull * h->num_layers * h->nchunks;
  if (h->off_table != LMC_PACK_HEADER_BYTES || h->off_static != lmc_r16_64(h->off_table + 8 * (N + 1)) ||
      h->off_streams != h->off_static + (uint64_t)h->nchunks * h->static_stride || h->total_bytes > nbytes ||
      h->total_bytes < h->off_streams)
    return false;
  const uint64_t* t = (const uint64_t*)(b + h->off_table);
  if (t[0] != 0 || t[N] != h->total_bytes - h->off_streams) return false;
  for (uint64_t i = 0; i < N; i++)
    if (t[i + 1] < t[i] || (t[i] & 15)) return false;
  return true;
}

int lmc_pack_info(const void* pack_h, uint64_t nbytes, lmc_pack_header* out) {
  lmc_pack_header h;
  if (!out || !pack_ok((const u8*)pack_h, nbytes, &h)) return LMC_ERR_INVALID;
  *out = h;
  return LMC_OK;
}

int lmc_pack_extract(const void* pack_h, uint64_t nbytes, int32_t chunk, void* blob_out, uint64_t cap, uint32_t* size_out) {
  lmc_pack_header h;
  const u8* b = (const u8*)pack_h;
  if (!pack_ok(b, nbytes, &h) || chunk < 0 || (uint32_t)chunk >= h.nchunks || !blob_out || !size_out) return LMC_ERR_INVALID;
  const u8* st = b + h.off_static + (uint64_t)chunk * h.static_stride;
  lmc_blob_header bh;
  memcpy(&bh, st, sizeof bh);
  // the static slot is the head of the blob: its header says how long the head and the whole blob are
  if (bh.magic != LMC_BLOB_MAGIC || bh.off_streams > h.static_stride || bh.off_streams < sizeof bh ||
      bh.total_bytes != bh.off_streams + bh.stream_bytes || bh.total_bytes > cap || bh.num_layers != h.num_layers)
    return LMC_ERR_INVALID;
  memcpy(blob_out, st, bh.off_streams);
  const uint64_t* t = (const uint64_t*)(b + h.off_table);
  const uint32_t L = h.num_layers, n = h.nchunks;
  uint64_t at = bh.off_streams;
  for (uint32_t p = 0; p < 2 * L; p++) {  // plane order of the blob and of the pack: K planes of every layer, then V planes
    const uint64_t i = (uint64_t)p * n + (uint32_t)chunk;
    const uint64_t len = t[i + 1] - t[i];
    if (at + len > bh.total_bytes) return LMC_ERR_INVALID;
    memcpy((u8*)blob_out + at, b + h.off_streams + t[i], len);
    at += len;
  }
  if (at != bh.total_bytes) return LMC_ERR_INVALID;
  *size_out = bh.total_bytes;
  return LMC_OK;
}

int lmc_load_pack(lmc_ctx* c, const void* pack_h, uint64_t pack_bytes, int32_t chunk_begin, int32_t nchunks,
                  const lmc_kv_layout* dst, int32_t dst_tok0, int32_t layers_per_range, lmc_event_t* range_events,
                  uint32_t* job_status, lmc_stream_t stream) {
  lmc_pack_header h;
  if (!c || !layout_ok(dst, false) || layers_per_range < 0 || nchunks < 0 || chunk_begin < 0 ||
      !pack_ok((const u8*)pack_h, pack_bytes, &h))
    return LMC_ERR_INVALID;
  const int L = dst->num_layers, H = dst->num_heads, D = dst->head_size;
  if ((uint32_t)L != h.num_layers || (uint32_t)H != h.num_heads || (uint32_t)D != h.head_size || H * D > LMC_MAX_CHANNELS ||
      (uint32_t)chunk_begin >= h.nchunks || (uint32_t)nchunks > h.nchunks - (uint32_t)chunk_begin)
    return LMC_ERR_INVALID;
  const int n = (int)h.nchunks, c0 = chunk_begin, m = nchunks ? nchunks : n - c0;  // chunks c0 .. c0 + m of the pack
  const u8* b = (const u8*)pack_h;
  const uint64_t* t = (const uint64_t*)(b + h.off_table);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const int step = layers_per_range > 0 && layers_per_range < L ? layers_per_range : L;
  std::lock_guard<std::mutex> lk(c->mu);
  if ((rc = legs_init(c))) return rc;
  if (c->load_bytes < h.total_bytes) {
    if (c->load_used) HIP_TRY(hipEventSynchronize(c->load_free));
    if ((rc = ws_grow((void**)&c->load_slots, &c->load_bytes, (size_t)h.total_bytes))) return rc;
  }
  hipStream_t cs = c->copy_stream;  // ONE queue: the ranges must arrive in layer order
  if (c->load_used) HIP_TRY(hipStreamWaitEvent(cs, c->load_free, 0));  // the previous load's decodes have read the buffer
  // the device copy keeps the pack's offsets: table, static slots and segments land where they lie in the pack
  u8* dev = c->load_slots;
  HIP_TRY(hipMemcpyAsync(dev + h.off_table, b + h.off_table, 8 * (2 * (size_t)L * n + 1), hipMemcpyHostToDevice, cs));
  HIP_TRY(hipMemcpyAsync(dev + h.off_static + (size_t)c0 * h.static_stride, b + h.off_static + (size_t)c0 * h.static_stride,
                         (size_t)m * h.static_stride, hipMemcpyHostToDevice, cs));
  DecodeArgs da;
  memset(&da, 0, sizeof da);
  if ((rc = decode_common(c, dev + h.off_static + (size_t)c0 * h.static_stride, h.static_stride, m, L, H, D, job_status, da)))
    return rc;
  da.dst = to_addr(dst); da.dst_tok0 = dst_tok0; da.chunk_tokens = (int)h.chunk_tokens;
  da.seg_off = (const unsigned long long*)(dev + h.off_table) + c0; da.seg_streams = dev + h.off_streams; da.seg_n = n;
  int r = 0;
  for (int l0 = 0; l0 < L; l0 += step, r++) {
    const int nl = l0 + step <= L ? step : L - l0;
    // the streams of layers l0 .. l0 + nl: one contiguous region when the whole pack is wanted, else one run of the
    // m chunks per (layer, kv)
    // (pack v3, plane order: the K planes of the range are one contiguous region, its V planes another)
    for (int kv = 0; kv < 2; kv++) {
      const int p0 = kv * L + l0;
      if (m == n) {
        const uint64_t lo = t[(uint64_t)p0 * n], hi = t[(uint64_t)(p0 + nl) * n];
        if (hi > lo) HIP_TRY(hipMemcpyAsync(dev + h.off_streams + lo, b + h.off_streams + lo, hi - lo, hipMemcpyHostToDevice, cs));
      } else {
        for (int p = p0; p < p0 + nl; p++) {
          const uint64_t lo = t[(uint64_t)p * n + c0], hi = t[(uint64_t)p * n + c0 + m];
          if (hi > lo) HIP_TRY(hipMemcpyAsync(dev + h.off_streams + lo, b + h.off_streams + lo, hi - lo, hipMemcpyHostToDevice, cs));
        }
      }
    }
    hipEvent_t ev;
    if ((rc = next_event(c, &ev))) return rc;
    HIP_TRY(hipEventRecord(ev, cs));
    HIP_TRY(hipStreamWaitEvent(s, ev, 0));
    da.layer_begin = l0; da.layer_count = nl;
    const long long nstreams = (long long)m * 2 * nl * da.G;
    const dim3 grid((unsigned)((nstreams + DEC_WAVES - 1) / DEC_WAVES));
    const bool paged = dst->slot_mapping != nullptr;
    if (dst->dtype == LMC_DTYPE_BF16) {
      if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, true>), grid, dim3(64 * DEC_WAVES), 0, s, da);
      else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_BF16, false>), grid, dim3(64 * DEC_WAVES), 0, s, da);
    } else {
      if (paged) hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, true>), grid, dim3(64 * DEC_WAVES), 0, s, da);
      else hipLaunchKernelGGL((k_decode<false, LMC_DTYPE_FP16, false>), grid, dim3(64 * DEC_WAVES), 0, s, da);
    }
    HIP_TRY(hipGetLastError());
    if (range_events && range_events[r]) HIP_TRY(hipEventRecord((hipEvent_t)range_events[r], s));
  }
  HIP_TRY(hipEventRecord(c->load_free, s));
  c->load_used = true;
  return LMC_OK;
}

int lmc_blob_info(const void* blob_h, size_t nbytes, lmc_blob_header* out) {
  if (!blob_h || !out || nbytes < sizeof(lmc_blob_header)) return LMC_ERR_INVALID;
  lmc_blob_header h;
  memcpy(&h, blob_h, sizeof h);
  if (h.magic != LMC_BLOB_MAGIC || h.version != LMC_BLOB_VERSION || h.header_bytes != LMC_HEADER_BYTES) return LMC_ERR_INVALID;
  if (h.num_layers == 0 || h.ntokens == 0 || h.num_heads == 0 || h.head_size == 0) return LMC_ERR_INVALID;
  if (h.num_layers > LMC_MAX_PLANES / 2 || h.ntokens > 65535u || !lmc_model_valid(h.model, h.ntokens)) return LMC_ERR_INVALID;
  if ((uint64_t)h.num_heads * h.head_size > LMC_MAX_CHANNELS) return LMC_ERR_INVALID;
  lmc_blob_header ref;
  memset(&ref, 0, sizeof ref);
  lmc_blob_layout(h.num_layers, h.ntokens, h.num_heads, h.head_size, &ref);
  if (h.nchannels != ref.nchannels || h.nplanes != ref.nplanes || h.ngroups != ref.ngroups || h.lp != ref.lp ||
      h.off_bins != ref.off_bins || h.off_scales != ref.off_scales || h.off_scsum != ref.off_scsum ||
      h.off_gdir != ref.off_gdir || h.off_streams != ref.off_streams)
    return LMC_ERR_INVALID;
  if (h.total_bytes != h.off_streams + h.stream_bytes || h.total_bytes > nbytes) return LMC_ERR_INVALID;
  *out = h;
  return LMC_OK;
}

}  // extern "C"
