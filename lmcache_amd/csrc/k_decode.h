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
// One wave = one group stream, lane = channel.  Per token: 5-step branchless
// binary search of the lane's CDF column in LDS ([entry][lane] u16, bank =
// lane/2: conflict free), state update, ballot/mbcnt pop of 16-bit words from
// the tail of the stream.
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

#define DEC_WAVE_BYTES 8448  // cdfT 32*64*2 = 4096 | stage / scales 4224 | lut 128

template <bool SYMOUT, int DT_OUT>
__global__ __launch_bounds__(256) void k_decode(DecodeArgs a) {
  __shared__ __attribute__((aligned(16))) u8 lds_all[4 * DEC_WAVE_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gid = (long long)blockIdx.x * 4 + wave;
  const int n = a.P * a.G;
  if (gid >= (long long)a.nchunks * n) return;
  u8* wl = lds_all + wave * DEC_WAVE_BYTES;
  u16* cdfT = reinterpret_cast<u16*>(wl);              // [32][64]
  u16* stage = reinterpret_cast<u16*>(wl + 4096);      // [64][33] raw rows, later float sc[T<=1056]
  float* sc = reinterpret_cast<float*>(wl + 4096);
  float* lut = reinterpret_cast<float*>(wl + 4096 + 4224);  // [32]

  const int chunk = (int)(gid / n);
  const int pg = (int)(gid - (long long)chunk * n);
  const int p = pg / a.G, g = pg - p * a.G;
  const u8* blob = a.blobs + (long long)chunk * a.blob_stride;
  const u32* hd = reinterpret_cast<const u32*>(blob);
  const u32 T = hd[4];
  const u32 src_dtype = hd[2];
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G);
  if (hd[0] != LMC_BLOB_MAGIC || hd[7] != (u32)a.C || hd[8] != (u32)a.P || hd[15] != bo.streams) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_HEADER);
    return;
  }
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  // ---- CDF rows of this group -> LDS, transposed to [entry][lane] ----------
  {
    const int nvalid = min(64, a.C - g * 64);
    const int n16 = nvalid * (LMC_LP * 2) / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(blob + bo.cdf + ((long long)p * a.C + g * 64) * (LMC_LP * 2));
    uint4* d4 = reinterpret_cast<uint4*>(stage);
    for (int i = lane; i < n16; i += 64) d4[i] = s4[i];
  }
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < 32; i++) cdfT[i * 64 + lane] = active ? stage[lane * LMC_LP + i] : (u16)i;
  wave_lds_fence();

  // ---- per-token scales and the dequantisation LUT -------------------------
  const u16* scl = reinterpret_cast<const u16*>(blob + bo.scales) + (long long)p * T;
  const bool sc_in_lds = !SYMOUT && T <= 1056u;
  if (sc_in_lds)
    for (u32 t = lane; t < T; t += 64) sc[t] = h2f_rt(scl[t], (int)src_dtype);
  if (!SYMOUT && lane < 32) {
    const float Cf = (float)((int)blob[bo.bins + p] / 2 - 1);
    float v = (float)lane - Cf;
    lut[lane] = v / Cf;
  }
  wave_lds_fence();

  // ---- stream ----------------------------------------------------------------
  const u32* gend = reinterpret_cast<const u32*>(blob + bo.gend);
  const u32 end = gend[pg];
  const u32 start = pg == 0 ? 0u : ((gend[pg - 1] + 15u) & ~15u);
  const u16* words = reinterpret_cast<const u16*>(blob + bo.streams + start);
  const u32 nw = (end - start) >> 1;
  if (end < start + 256u || bo.streams + end > hd[17]) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
    return;
  }
  u32 w = nw - 128u;
  u32 x = (u32)words[w + 2 * lane] | ((u32)words[w + 2 * lane + 1] << 16);

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
    u32 s = 0, lo = 0, hi = 65536u;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1) {
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
    if (mask) {
      const u32 cnt = (u32)__popcll(mask);
      if (cnt > w) { bad = true; break; }
      w -= cnt;
      if (need) x = (x << 16) | (u32)words[w + lane_rank(mask)];
    }
    if (SYMOUT) {
      if (active) sbase[(long long)t * a.C] = (int8_t)s;
    } else {
      const int td = tdst0 + (int)t;
      if (td >= 0 && active) {
        const float scale = sc_in_lds ? sc[t] : h2f_rt(scl[t], (int)src_dtype);
        const float val = lut[s] * scale;
        const u32 bits = DT_OUT == LMC_DTYPE_BF16 ? f2bf16(val) : f2fp16(val);
        dbase[lmc_tok_off(a.dst, td)] = (u16)bits;
      }
    }
  }
  const bool state_bad = active && x != LMC_RANS_L;
  if (bad || w != 0 || __ballot(state_bad)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
  }
}
