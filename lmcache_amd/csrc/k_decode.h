// k_decode.h -- per-(plane, 64-channel group) interleaved rANS decode fused with
// dequantise, 16-bit cast and the scatter into the caller's KV layout
// (SURVEY.md section 8a rows a13-a16, a18, a19).
//
// Replaces decode_chunk + torchac_cuda.decode_fast_prefsum
// (lmcache/storage_backend/serde/cachegen_decoder.py:51-66), the uint8->fp32
// inflation (:95-104), do_dequantize (:24-35), stack/reshape/permute/.to(16-bit)
// (:177-200) and the engine's torch.cat over chunks (cache_engine.py:362-368):
// the decoded value of (plane, token, channel) is written exactly once, in its
// final place.
//
//   t = ((q - C) / C) * max1      three rounded fp32 ops; (q-C)/C comes from a
//                                  32-entry per-plane LUT built with the same
//                                  fp32 subtraction and IEEE division
//   out = RNE16(t)
//
// One wave = one group stream, lane = channel.  Per token: first search level
// on three register pivots, then three dependent probes of the lane's CDF
// column in LDS ([entry][lane] u16, bank = lane/2: conflict free), state
// update, ballot/mbcnt pop of 16-bit words.  The words come from a 512-word LDS
// ring refilled half a ring ahead of the consumer with coalesced loads, so no
// global-memory latency sits on the per-token dependency chain.  6.1 KiB of
// LDS per wave -> 6 waves per SIMD.
#pragma once
#include "lmc_device.h"

struct DecodeArgs {
  const u8* blobs;
  long long blob_stride;
  int nchunks;
  KvAddr dst;          // SYMOUT=false
  int dst_tok0, chunk_tokens;
  int8_t* sym_out;     // SYMOUT=true: [P][T][C] (single blob)
  int P, C, G;
  u32* status;
};

// per-wave LDS: cdfT [32][64] u16 (4096) | word ring 512 x u16 (1024) | scales 512 x u16 (1024) | lut 32 x f32 (128)
#define DEC_WAVE_BYTES 6272
#define DEC_RING_WORDS 512
#define DEC_SCALE_TOKENS 512

template <bool SYMOUT, int DT_OUT>
__global__ __launch_bounds__(256) void k_decode(DecodeArgs a) {
  __shared__ __attribute__((aligned(16))) u8 lds_all[4 * DEC_WAVE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform -> SGPRs
  const long long gid = (long long)blockIdx.x * 4 + wave;
  const int n = a.P * a.G;
  if (gid >= (long long)a.nchunks * n) return;
  u8* wl = lds_all + wave * DEC_WAVE_BYTES;
  u16* cdfT = reinterpret_cast<u16*>(wl);                 // [32][64]: entry-major, bank = lane/2
  u16* ring = reinterpret_cast<u16*>(wl + 4096);          // stream words, indexed by consumption order
  u16* scs = reinterpret_cast<u16*>(wl + 4096 + 1024);    // raw per-token scales of this plane
  float* lut = reinterpret_cast<float*>(wl + 4096 + 2048);  // (q - C) / C

  const int chunk = (int)(gid / n);
  const int pg = (int)(gid - (long long)chunk * n);
  const int p = pg / a.G, g = pg - p * a.G;
  const u8* blob = a.blobs + (long long)chunk * a.blob_stride;
  const u32* hd = reinterpret_cast<const u32*>(blob);
  const u32 T = __builtin_amdgcn_readfirstlane(hd[4]);
  const u32 src_dtype = __builtin_amdgcn_readfirstlane(hd[2]);
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G);
  if (hd[0] != LMC_BLOB_MAGIC || hd[7] != (u32)a.C || hd[8] != (u32)a.P || hd[15] != bo.streams) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_HEADER);
    return;
  }
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  // ---- CDF rows of this group -> LDS [entry][lane] (coalesced 2-byte loads, transposed LDS writes) ----
  {
    const u32 total = (u32)min(64, a.C - g * 64) * LMC_LP;
    const u16* src = reinterpret_cast<const u16*>(blob + bo.cdf) + ((long long)p * a.C + g * 64) * LMC_LP;
#pragma unroll 1
    for (u32 e0 = 0; e0 < total; e0 += 64 * 11) {  // 33 = 3 x 11 sweeps; 11 loads in flight
      u16 v[11];
#pragma unroll
      for (int i = 0; i < 11; i++) {
        const u32 e = e0 + i * 64 + lane;
        v[i] = e < total ? src[e] : (u16)0;
      }
#pragma unroll
      for (int i = 0; i < 11; i++) {
        const u32 e = e0 + i * 64 + lane;
        const u32 cl = div33(e), s = e - cl * LMC_LP;
        if (e < total && s < 32u) cdfT[s * 64 + cl] = v[i];
      }
    }
    if (!active) {  // idle lanes: any strictly increasing column keeps the search in range
#pragma unroll
      for (int i = 0; i < 32; i++) cdfT[i * 64 + lane] = (u16)i;
    }
  }
  // ---- per-token scales and the dequantisation LUT -------------------------
  const u16* scl = reinterpret_cast<const u16*>(blob + bo.scales) + (long long)p * T;
  const bool sc_in_lds = !SYMOUT && T <= (u32)DEC_SCALE_TOKENS;
  if (sc_in_lds)
    for (u32 t = lane; t < T; t += 64) scs[t] = scl[t];
  if (!SYMOUT && lane < 32) {
    const float Cf = (float)((int)blob[bo.bins + p] / 2 - 1);
    const float v = (float)lane - Cf;
    lut[lane] = v / Cf;
  }

  // ---- stream ----------------------------------------------------------------
  const u32* gend = reinterpret_cast<const u32*>(blob + bo.gend);
  const u32 end = __builtin_amdgcn_readfirstlane(gend[pg]);
  const u32 start = pg == 0 ? 0u : ((__builtin_amdgcn_readfirstlane(gend[pg - 1]) + 15u) & ~15u);
  const u16* words = reinterpret_cast<const u16*>(blob + bo.streams + start);
  if (end < start + 256u || bo.streams + end > hd[17]) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
    return;
  }
  const u32 nwords = ((end - start) >> 1) - 128u;  // 16-bit words in front of the 64 states
  u32 x = (u32)words[nwords + 2 * lane] | ((u32)words[nwords + 2 * lane + 1] << 16);
  // Word k (k = 0, 1, ... in the order the decoder consumes them) is words[nwords - 1 - k]; it is staged
  // in ring[k % 512].  The ring is refilled half a ring (256 words) at a time: the loads of the next half
  // are ISSUED into registers when fewer than 384 words are ahead of the consumer and WRITTEN to the ring
  // when fewer than 256 are (by then the older half is fully consumed) -- ~10 tokens later, so their
  // latency is off the per-token dependency chain.
  u16 pend[4];
  auto ring_issue = [&](u32 k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32 k = k0 + i * 64 + lane;
      pend[i] = k < nwords ? words[nwords - 1 - k] : (u16)0;
    }
  };
  auto ring_commit = [&](u32 k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) ring[(k0 + i * 64 + lane) & (DEC_RING_WORDS - 1)] = pend[i];
  };
  ring_issue(0);
  ring_commit(0);
  ring_issue(256);
  ring_commit(256);
  u32 filled = 512;        // words [0, filled) are in the ring
  bool pending = false;    // loads for [filled, filled + 256) are in flight (wave-uniform)
  u32 consumed = 0;        // wave-uniform
  wave_lds_fence();

  // pivots of the first search level live in registers
  const u32 p8 = cdfT[8 * 64 + lane], p16 = cdfT[16 * 64 + lane], p24 = cdfT[24 * 64 + lane];

  // destination addressing (row independent part)
  u16* dbase = nullptr;
  int8_t* sbase = nullptr;
  if (SYMOUT) {
    sbase = a.sym_out + (long long)p * T * a.C + c;
  } else {
    const int h = c / a.dst.D, d = c - h * a.dst.D;
    dbase = const_cast<u16*>(lmc_plane_base(a.dst, p)) + (long long)h * a.dst.stride_head + d;
  }
  const int tdst0 = a.dst_tok0 + chunk * a.chunk_tokens;
  bool bad = false;

  for (u32 t = 0; t < T; t++) {
    const u32 slot = x & 0xffffu;
    // level 1: quadrant from register pivots; levels 2-4: three dependent LDS probes
    const bool g8 = p8 <= slot, g16 = p16 <= slot, g24 = p24 <= slot;
    u32 s = g24 ? 24u : g16 ? 16u : g8 ? 8u : 0u;
    u32 lo = g24 ? p24 : g16 ? p16 : g8 ? p8 : 0u;
    u32 hi = g24 ? 65536u : g16 ? p24 : g8 ? p16 : p8;
#pragma unroll
    for (int step = 4; step >= 1; step >>= 1) {
      const u32 v = cdfT[(s + step) * 64 + lane];
      const bool ge = v <= slot;
      s = ge ? s + step : s;
      lo = ge ? v : lo;
      hi = ge ? hi : v;
    }
    const u32 f = hi - lo;
    x = __umul24(f, x >> 16) + slot - lo;
    const bool need = active && (x < LMC_RANS_L);
    const u64 mask = __ballot(need);
    const u32 cnt = (u32)__popcll(mask);
    if (consumed + cnt > nwords) { bad = true; break; }
    // the encoder appended this token's words in ascending lane order; counted from the tail that is
    // descending, so rank r of cnt takes consumption index consumed + cnt - 1 - r
    if (need) x = (x << 16) | (u32)ring[(consumed + cnt - 1u - lane_rank(mask)) & (DEC_RING_WORDS - 1)];
    consumed += cnt;
    if (!pending && consumed + 384u >= filled && filled < nwords) {
      ring_issue(filled);
      pending = true;
    }
    if (pending && consumed + 256u >= filled) {
      wave_lds_fence();  // every lane's reads of the older half are done
      ring_commit(filled);
      filled += 256;
      pending = false;
      wave_lds_fence();
    }
    if (SYMOUT) {
      if (active) sbase[(long long)t * a.C] = (int8_t)s;
    } else {
      const int td = tdst0 + (int)t;
      if (td >= 0 && active) {
        const float scale = h2f_rt(sc_in_lds ? scs[t] : scl[t], (int)src_dtype);
        const float val = lut[s] * scale;
        const u32 bits = DT_OUT == LMC_DTYPE_BF16 ? f2bf16(val) : f2fp16(val);
        dbase[lmc_tok_off(a.dst, td)] = (u16)bits;
      }
    }
  }
  const bool state_bad = active && x != LMC_RANS_L;
  if (bad || consumed != nwords || __ballot(state_bad)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
  }
}
