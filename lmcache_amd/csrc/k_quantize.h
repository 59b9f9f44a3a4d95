// k_quantize.h -- fused |x| row-max + quantise.
//
// Replaces _split_kv + torch_quant_vectorized + torch.cat
// (lmcache/storage_backend/serde/cachegen_encoder.py:40-61, 76-91, 278-285):
//   MAX = bins//2 - 1;  max1 = amax(|x|, channels);  factor = MAX / max1
//   q = round(x * factor + MAX).to(int8)
// in fp32 with separately rounded mul and add (-ffp-contract=off), IEEE
// division, round-half-even.
//
// Mapping: a "row quad" = 4 consecutive tokens of one plane.  G lanes of a
// wave own one row quad (G = 16/32/64 by channel count), each lane owns NITER
// runs of 8 consecutive channels (one 16-byte load per row per run), so the
// four rows of a quad give 4*NITER independent 16-byte loads in flight per
// lane.  The row max is a packed-u16 integer max on |x| bit patterns
// (NaN > inf > finite, which is torch.amax's NaN propagation) reduced with
// xor-shuffles inside the G-lane group.
//
// Output, QUAD=true (pipeline): sym4[chunk][plane][quad][channel] u32 whose
// byte k is the symbol of token 4*quad+k -- a lane-local byte transpose, so
// the encoder later reads 4 tokens of its channel with one coalesced dword.
// Output, QUAD=false (lmc_quantize parity entry): int8 [P][T][C].
#pragma once
#include "lmc_device.h"

struct QuantArgs {
  KvAddr src;
  BinsArg bins;
  int tok_begin, tok_end, chunk_tokens, nchunks;
  int P, C, TQ;          // TQ = ceil(chunk_tokens / 4)
  long long nquads;      // nchunks * P * TQ
  u32* sym4;             // QUAD: [nchunks][P][TQ][C]
  int8_t* sym8;          // !QUAD: [P][T][C]
  u8* scale_base;        // scale of (chunk, p, t) at scale_base + chunk*scale_stride + 2*(p*Tc + t)
  long long scale_stride;
};

// Regular rows (finite, non-zero max): |x * factor| <= MAX(1 + 2^-23), so the rounded value is 0 .. 2*MAX.
__device__ __forceinline__ u32 quant_fast(float x, float factor, float maxf) {
  float y = x * factor;  // rounded
  float z = y + maxf;    // rounded separately (no FMA: -ffp-contract=off)
  return (u32)(int)__builtin_rintf(z);
}
// Special rows (max is 0 / denormal-tiny / inf / NaN): follow the reference's CPU float->int8
// conversion: NaN and |r| >= 2^31 give INT_MIN whose low byte is 0.
__device__ __forceinline__ u32 quant_special(float x, float factor, float maxf) {
  float y = x * factor;
  float z = y + maxf;
  float r = __builtin_rintf(z);
  return (__builtin_fabsf(r) < 2147483648.0f) ? ((u32)(int)r & 0xffu) : 0u;
}

template <int G, int NITER, int DT, bool QUAD>
__global__ __launch_bounds__(256) void k_quantize(QuantArgs a) {
  constexpr int RPW = LMC_WAVE / G;  // row quads per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, sl = lane % G;
  long long qid = ((long long)blockIdx.x * 4 + wave) * RPW + sub;
  const bool qvalid = qid < a.nquads;
  if (!qvalid) qid = a.nquads - 1;
  const int q = (int)(qid % a.TQ);
  const long long pc = qid / a.TQ;
  const int p = (int)(pc % a.P);
  const int chunk = (int)(pc / a.P);
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);

  // per-lane channel runs (row independent)
  long long coff[NITER];
  int c0[NITER];
  bool cval[NITER];
#pragma unroll
  for (int it = 0; it < NITER; it++) {
    c0[it] = (it * G + sl) * 8;
    cval[it] = c0[it] < a.C;
    int h = c0[it] / a.src.D, d = c0[it] - h * a.src.D;
    coff[it] = (long long)h * a.src.stride_head + d;
  }

  const u16* pbase = lmc_plane_base(a.src, p);
  uint4 v[4][NITER];
  bool tv[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int t = q * 4 + r;
    tv[r] = qvalid && t < Tc;
    const u16* rowp = pbase + (tv[r] ? lmc_tok_off(a.src, tok0 + t) : 0);
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      if (tv[r] && cval[it]) v[r][it] = ld_global_u4(rowp + coff[it]);
      else v[r][it] = make_uint4(0, 0, 0, 0);
    }
  }

  // row |x| max on bit patterns
  u32 mrow[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    u32 m = 0;
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      m = pk_max_u16(m, v[r][it].x & 0x7fff7fffu);
      m = pk_max_u16(m, v[r][it].y & 0x7fff7fffu);
      m = pk_max_u16(m, v[r][it].z & 0x7fff7fffu);
      m = pk_max_u16(m, v[r][it].w & 0x7fff7fffu);
    }
    mrow[r] = max(m & 0xffffu, m >> 16);
  }
  u32 m01 = mrow[0] | (mrow[1] << 16), m23 = mrow[2] | (mrow[3] << 16);
#pragma unroll
  for (int off = 1; off < G; off <<= 1) {
    m01 = pk_max_u16(m01, (u32)__shfl_xor((int)m01, off));
    m23 = pk_max_u16(m23, (u32)__shfl_xor((int)m23, off));
  }
  mrow[0] = m01 & 0xffffu; mrow[1] = m01 >> 16; mrow[2] = m23 & 0xffffu; mrow[3] = m23 >> 16;

  const float maxf = (float)((int)a.bins.b[p] / 2 - 1);

  // scales: first lane of the group
  if (sl == 0) {
    u16* sp = reinterpret_cast<u16*>(a.scale_base + (long long)chunk * a.scale_stride) + ((long long)p * Tc + q * 4);
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (tv[r]) sp[r] = (u16)mrow[r];
  }

  float factor[4];
  bool special[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float sf = h2f_rt(mrow[r], DT);
    factor[r] = maxf / sf;  // IEEE fp32 division
    special[r] = !(__builtin_fabsf(factor[r]) < __builtin_inff()) || !(sf < __builtin_inff());
  }

#pragma unroll
  for (int it = 0; it < NITER; it++) {
    u32 sy[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sy[r][2 * k] = quant_fast(h_lo<DT>(w[k]), factor[r], maxf);
        sy[r][2 * k + 1] = quant_fast(h_hi<DT>(w[k]), factor[r], maxf);
      }
      if (__ballot(special[r])) {  // wave-uniform and rare: redo this row's lanes the careful way
        if (special[r]) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            sy[r][2 * k] = quant_special(h_lo<DT>(w[k]), factor[r], maxf);
            sy[r][2 * k + 1] = quant_special(h_hi<DT>(w[k]), factor[r], maxf);
          }
        }
      }
    }
    if (!cval[it] || !qvalid) continue;
    if (QUAD) {
      u32 o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        u32 d = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) d |= (tv[r] ? sy[r][e] : 0u) << (8 * r);
        o[e] = d;
      }
      u32* dst = a.sym4 + (((long long)chunk * a.P + p) * a.TQ + q) * a.C + c0[it];
      *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[4], o[5], o[6], o[7]);  // kernel-arg pointers: global already
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (!tv[r]) continue;
        u32 lo = sy[r][0] | (sy[r][1] << 8) | (sy[r][2] << 16) | (sy[r][3] << 24);
        u32 hi = sy[r][4] | (sy[r][5] << 8) | (sy[r][6] << 16) | (sy[r][7] << 24);
        int8_t* dst = a.sym8 + ((long long)p * Tc + (q * 4 + r)) * a.C + c0[it];
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      }
    }
  }
}
