// k_quantize.h -- fused |x| row-max + quantise.
//
// Replaces _split_kv + torch_quant_vectorized + torch.cat
// (lmcache/storage_backend/serde/cachegen_encoder.py:40-61, 76-91, 278-285):
//   MAX = bins//2 - 1;  max1 = amax(|x|, channels);  factor = MAX / max1
//   q = round(x * factor + MAX).to(int8)
// in fp32 with separately rounded mul and add (-ffp-contract=off), IEEE
// division, round-half-even.
//
// Mapping: a task = one "row oct", 8 consecutive tokens of one plane, taken by the G lanes of a wave
// (G = 16/32/64 by channel count); each lane owns NITER runs of 8 consecutive channels (one 16-byte load
// per row per run), so a task keeps 8*NITER independent 16-byte loads in flight per lane (NITER <= 2).
// The row max is a packed-u16 integer max on |x| bit patterns (NaN > inf > finite, which is torch.amax's
// NaN propagation) reduced with xor-shuffles inside the G-lane group.
//
// Output, QUAD=true (pipeline): the symbol workspace of the coder, one region of TQ*C dwords per
// (chunk, plane), written with a lane-local transpose so that the coder later reads several tokens of its
// channel with one coalesced dword:
//   planes with > 17 bins  [quad][channel] u32, byte k = symbol of token 4*quad + k
//   planes with <= 17 bins [oct][channel] u32,  byte k = token 8*oct + k | token 8*oct + 4 + k << 4
// Output, QUAD=false (lmc_quantize parity entry): int8 [P][T][C].
#pragma once
#include "lmc_device.h"
#include "k_hist.h"

struct QuantArgs {
  KvAddr src;
  BinsArg bins;
  int tok_begin, tok_end, chunk_tokens, nchunks;
  int P, C, TQ;          // TQ = ceil(chunk_tokens / 4)
  long long sym_stride;  // QUAD: dwords between the workspace regions of consecutive (chunk, plane) pairs (>= TQ * C)
  int pc_limit;          // plane-chunks (chunk * P + plane) to quantise: nchunks * P
  u32* sym4;             // QUAD: symbol workspace, TQ*C dwords per (chunk, plane)
  int8_t* sym8;          // !QUAD: [P][T][C]
  u8* scale_base;        // scale of (chunk, p, t) at scale_base + chunk*scale_stride + 2*(p*Tc + t)
  long long scale_stride;
  unsigned long long* agg;  // the coder's look-back granules, zeroed here (saves a memset dispatch), or NULL
  long long agg_n;
  u32* sizes;               // ... and the job's size words [nchunks] (0 = "this chunk's encode did not finish"), or NULL
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// Regular rows (finite, non-zero max): |x * factor| <= MAX(1 + 2^-23), so z = x*factor + MAX lies in
// [-MAX 2^-23, 2 MAX + ...] and round-half-even gives 0 .. 2*MAX.  Two elements per instruction with the
// packed fp32 multiply and add (each rounded on its own: no FMA), and v_cvt_pk_u8_f32 does the RNE
// conversion, the clamp at 0 (-0.x -> 0) and the byte insertion in one op (probed on gfx950:
// tools/probes/cvt_pk_u8.hip -- RNE, saturating, NaN -> 0).
__device__ __forceinline__ f32x2_t quant_z2(float xlo, float xhi, f32x2_t factor2, f32x2_t maxf2) {
  f32x2_t x = {xlo, xhi};
  f32x2_t y = x * factor2;  // v_pk_mul_f32, rounded
  return y + maxf2;         // v_pk_add_f32, rounded separately (-ffp-contract=off)
}
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

// One task = 4 (row quad) or, NIB, 8 consecutive tokens starting at token t_first of plane p of one chunk,
// by the G lanes sl = 0..G-1.
//   sym_out     QUAD: the task's dwords, [channel]      sym8_plane  !QUAD: &sym8[p][0][0]
//   scale_out   &scales[chunk][p][t_first]
// QUAD output dword of channel c: byte k = symbol of token t_first + k.  NIB (planes whose symbols fit four
// bits: bins <= 17): byte k = symbol of token t_first + k | symbol of token t_first + 4 + k << 4, so the
// workspace round trip of most planes is half a byte per element.
// ROWS (4 or 2): rows loaded and reduced together -- 4 keeps 4*NITER 16-byte loads in flight per lane,
// 2 halves the registers.
// SPLIT > 1 (planes of more than 1024 channels, workspace output): SPLIT waves of the workgroup share the task, wave
// `slice` takes channels [slice, slice + 1) * NITER * 512, and the row maxima meet in `xmax` (LDS, [8][SPLIT]) behind
// ONE workgroup barrier per task -- a wave then holds 8 rows x 1024 channels (98 VGPRs, 4 waves per SIMD) where a
// whole 4096-channel row per wave costs 241 VGPRs and 2 waves per SIMD (C = 4096: 3.5 -> 4.3 TB/s).
// HIST (the fused encode's narrow planes, round 6): the task's symbols also go into the work item's histogram (k_hist.h)
// -- hist_lane4 = 4 * the task's virtual quantising lane, hist_it its virtual channel run (wave-uniform: a wave pass
// quantises row octs of ONE plane).
template <int G, int NITER, int DT, bool QUAD, bool NIB, int ROWS = 4, int NQ = NIB ? 2 : 1, int SPLIT = 1, bool HIST = false>
__device__ __forceinline__ void quantize_task(const KvAddr& src, int p, int tok0, int Tc, int t_first, bool qvalid,
                                              bool q1valid, int C, float maxf, u32* sym_out, int8_t* sym8_plane,
                                              u16* scale_out, int sl, int slice = 0, u32* xmax = nullptr,
                                              u32 hist_lane4 = 0, int hist_it = 0) {
  static_assert(!HIST || (QUAD && NITER == 1 && NQ == 2 && SPLIT == 1), "the histogram rides on the fused kernel's narrow tasks");
  static_assert(SPLIT == 1 || (QUAD && ROWS == 4 * NQ), "a split task exchanges all its rows' maxima at once");
  static_assert(ROWS == 2 || ROWS == 4 || (ROWS == 8 && NQ == 2), "rows in flight: a power of two within the task");
  static_assert(QUAD || (!NIB && NQ == 1), "nibble packing and two-quad tasks are workspace formats");
  static_assert(!NIB || NQ == 2, "a nibble dword holds two row quads");
  // per-lane channel runs (row independent)
  long long coff[NITER];
  int c0[NITER];
  bool cval[NITER];
#pragma unroll
  for (int it = 0; it < NITER; it++) {
    c0[it] = ((slice * NITER + it) * G + sl) * 8;
    cval[it] = c0[it] < C;
    int h = c0[it] / src.D, d = c0[it] - h * src.D;
    coff[it] = (long long)h * src.stride_head + d;
  }
  const u16* pbase = lmc_plane_base(src, p);
  const f32x2_t maxf2 = {maxf, maxf};
  // QUAD: o[hq][it][e] byte r = symbol of (token t_first + 4 hq + r, channel c0[it] + e); bytes of tokens past
  // the end of a ragged chunk are never read by the coder, so they need no masking
  u32 o[NQ][NITER][8];
#pragma unroll
  for (int hq = 0; hq < NQ; hq++)
#pragma unroll
    for (int it = 0; it < NITER; it++)
#pragma unroll
      for (int e = 0; e < 8; e++) o[hq][it][e] = 0;

#pragma unroll
  for (int r0 = 0; r0 < 4 * NQ; r0 += ROWS) {
    uint4 v[ROWS][NITER];
    bool tv[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      int t = t_first + r0 + r;
      tv[r] = qvalid && t < Tc;
      const u16* rowp = pbase + (tv[r] ? lmc_tok_off(src, tok0 + t) : 0);
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        if (tv[r] && cval[it]) v[r][it] = ld_global_u4(rowp + coff[it]);  // (non-temporal here: two-kernel path +0.5 %)
        else v[r][it] = make_uint4(0, 0, 0, 0);
      }
    }

    // row |x| max on bit patterns, two rows per register through the xor-shuffle reduction
    u32 mrow[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
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
#pragma unroll
    for (int r = 0; r < ROWS; r += 2) {
      u32 m2 = mrow[r] | (mrow[r + 1] << 16);
#pragma unroll
      for (int off = 1; off < G; off <<= 1) m2 = pk_max_u16(m2, (u32)__shfl_xor((int)m2, off));
      mrow[r] = m2 & 0xffffu;
      mrow[r + 1] = m2 >> 16;
    }

    if constexpr (SPLIT > 1) {  // the row maxima of the SPLIT channel slices meet in LDS
      if (sl == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) xmax[r * SPLIT + slice] = mrow[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        u32 m = 0;
#pragma unroll
        for (int k = 0; k < SPLIT; k++) m = max(m, xmax[r * SPLIT + k]);
        mrow[r] = m;
      }
    }
    // scales: first lane of the group (of the first slice)
    if (sl == 0 && slice == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; r++)
        if (tv[r]) scale_out[r0 + r] = (u16)mrow[r];
    }

    float factor[ROWS];
    bool special[ROWS];
    bool any_special = false;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      float sf = h2f_rt(mrow[r], DT);
      factor[r] = maxf / sf;  // IEEE fp32 division
      special[r] = !(__builtin_fabsf(factor[r]) < __builtin_inff()) || !(sf < __builtin_inff());
      any_special |= special[r];
    }
    const bool slow = __ballot(any_special) != 0;  // wave-uniform and rare (zero / inf / NaN rows, ragged tails)
    // one symbol at a time; special rows follow the CPU float -> int8 conversion of the reference
    auto sym_of = [&](int r, float x) -> u32 {
      return (special[r] ? quant_special(x, factor[r], maxf) : quant_fast(x, factor[r], maxf)) & 0xffu;
    };

#pragma unroll
    for (int it = 0; it < NITER; it++) {
      if (!cval[it] || !qvalid) continue;
      if (QUAD) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
          if (!slow) {  // four chained byte-inserting conversions per output dword
            const f32x2_t f2 = {factor[r], factor[r]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const f32x2_t z = quant_z2(h_lo<DT>(w[k]), h_hi<DT>(w[k]), f2, maxf2);
              o[(r0 + r) >> 2][it][2 * k] = __builtin_amdgcn_cvt_pk_u8_f32(z.x, (r0 + r) & 3, o[(r0 + r) >> 2][it][2 * k]);
              o[(r0 + r) >> 2][it][2 * k + 1] = __builtin_amdgcn_cvt_pk_u8_f32(z.y, (r0 + r) & 3, o[(r0 + r) >> 2][it][2 * k + 1]);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              o[(r0 + r) >> 2][it][2 * k] |= sym_of(r, h_lo<DT>(w[k])) << (8 * ((r0 + r) & 3));
              o[(r0 + r) >> 2][it][2 * k + 1] |= sym_of(r, h_hi<DT>(w[k])) << (8 * ((r0 + r) & 3));
            }
          }
        }
      } else {  // int8 [P][T][C] output of lmc_quantize
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          if (!tv[r]) continue;
          const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
          u32 lo = 0, hi = 0;
#pragma unroll
          for (int k = 0; k < 2; k++) {
            lo |= (sym_of(r, h_lo<DT>(w[k])) << (16 * k)) | (sym_of(r, h_hi<DT>(w[k])) << (16 * k + 8));
            hi |= (sym_of(r, h_lo<DT>(w[k + 2])) << (16 * k)) | (sym_of(r, h_hi<DT>(w[k + 2])) << (16 * k + 8));
          }
          int8_t* dst = sym8_plane + ((long long)(t_first + r0 + r)) * C + c0[it];
          *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
        }
      }
    }
  }
  if (QUAD && qvalid) {
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      if (!cval[it]) continue;
      u32* dst = sym_out + c0[it];  // kernel-arg pointers: global already
      if (NIB) {
        u32 w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = o[0][it][e] | (o[NQ - 1][it][e] << 4);
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(dst + 4) = make_uint4(w[4], w[5], w[6], w[7]);
        if constexpr (HIST) {
#pragma unroll
          for (int e = 0; e < 8; e++) o[0][it][e] = w[e];  // the histogram below counts the merged dwords
        }
      } else {
#pragma unroll
        for (int hq = 0; hq < NQ; hq++) {  // byte format: the task's row quads are adjacent [quad][channel] rows
          if (hq == 1 && !q1valid) continue;
          u32* d2 = dst + (long long)hq * C;
          *reinterpret_cast<uint4*>(d2) = make_uint4(o[hq][it][0], o[hq][it][1], o[hq][it][2], o[hq][it][3]);
          *reinterpret_cast<uint4*>(d2 + 4) = make_uint4(o[hq][it][4], o[hq][it][5], o[hq][it][6], o[hq][it][7]);
        }
      }
    }
  }
  if constexpr (HIST) {
    // rows of the oct that are tokens of the chunk (per lane group: the groups of a wave pass hold different octs); a byte
    // plane leaves token 0 out (its u8 counters must not reach 256: k_hist.h)
    const int nrow = (qvalid && cval[0]) ? min(8, Tc - t_first) : 0;
    const bool all8 = __ballot(nrow != 8) == 0;           // wave-uniform: every group holds a full oct (the usual case)
    const bool tok0_here = !NIB && __ballot(nrow > 0 && t_first == 0) != 0;  // wave-uniform: some group holds token 0
    u32 ad[2] = {hist_lane4, hist_lane4};
    auto add_all = [&](auto it_tag) {
      constexpr int IT = decltype(it_tag)::value;
      static_for<8>([&](auto e_tag) {
        constexpr int E = decltype(e_tag)::value;
        if (all8 && !tok0_here) {
          if constexpr (NIB) plane_hist_dword<true, IT, E, 0>(o[0][0][E], ad, [](int) { return true; });
          else {
            plane_hist_dword<false, IT, E, 0>(o[0][0][E], ad, [](int) { return true; });
            plane_hist_dword<false, IT, E, 1>(o[1][0][E], ad, [](int) { return true; });
          }
        } else {  // a partial or absent oct in some group, or token 0 of a byte plane: per-lane tests (exec-masked adds)
          auto counted = [&](int row) { return row < nrow && (NIB || t_first + row != 0); };
          if constexpr (NIB) plane_hist_dword<true, IT, E, 0>(o[0][0][E], ad, counted);
          else {
            plane_hist_dword<false, IT, E, 0>(o[0][0][E], ad, counted);
            plane_hist_dword<false, IT, E, 1>(o[1][0][E], ad, counted);
          }
        }
      });
    };
    if (hist_it) add_all(IntTag<1>{});  // (wave-uniform)
    else add_all(IntTag<0>{});
  }
}

// A wave (G = 64) or a G-lane group takes one row OCT: tokens 8*oct .. 8*oct + 7.  Planes with bins <= 17
// get the nibble-packed workspace format (one dword per channel for the oct), the others two row quads.
// grid = (ceil(TO / (4 * RPW)), P, chunks), TO = ceil(TQ / 2): plane and chunk come straight from the block
// index, and with G = 64 everything but the channel offset is wave-uniform and lives in SGPRs.
template <int G, int NITER, int DT, bool QUAD, int SPLIT = 1>
__global__ __launch_bounds__(256) void k_quantize(QuantArgs a) {
  static_assert(SPLIT == 1 || (QUAD && G == LMC_WAVE && (SPLIT == 2 || SPLIT == 4)), "split tasks: whole waves, workspace output");
  constexpr int RPW = LMC_WAVE / G;  // tasks per wave
  __shared__ u32 xmax[SPLIT > 1 ? (4 / SPLIT) * 8 * SPLIT : 1];  // row maxima of the split tasks of this workgroup
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int sub = lane / G, sl = lane % G;
  const int p = (int)blockIdx.y, chunk = (int)blockIdx.z;
  if (QUAD && a.agg) {  // the launch has at least 256 threads per plane-chunk, a plane-chunk at most 64 granules
    const long long i = (((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (i < a.agg_n) a.agg[i] = 0ull;
    if (a.sizes && i < a.nchunks) a.sizes[i] = 0u;
  }
  const int TO = (a.TQ + 1) >> 1;
  int oct = SPLIT > 1 ? ((int)blockIdx.x * 4 + wave) / SPLIT : ((int)blockIdx.x * 4 + wave) * RPW + sub;
  const int slice = SPLIT > 1 ? wave % SPLIT : 0;
  const bool ovalid = oct < TO && chunk * a.P + p < a.pc_limit;
  if (!ovalid) oct = 0;
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);
  const int bins = (int)a.bins.b[p];
  const float maxf = (float)(bins / 2 - 1);
  u16* scale_out = reinterpret_cast<u16*>(a.scale_base + (long long)chunk * a.scale_stride) + ((long long)p * Tc + oct * 8);
  u32* sym_pc = QUAD ? a.sym4 + ((long long)chunk * a.P + p) * a.sym_stride : nullptr;  // this plane-chunk's workspace
  int8_t* sym8_plane = QUAD ? nullptr : a.sym8 + (long long)p * Tc * a.C;
  if constexpr (QUAD) {
    constexpr int ROWS = NITER <= 2 ? 8 : 4;  // 16-byte loads in flight per lane: ROWS * NITER
    u32* const sym_out = sym_pc + (long long)oct * (lmc_sym_nibbles(bins) ? 1 : 2) * a.C;
    const bool q1valid = 2 * oct + 1 < a.TQ;
    u32* const xm = xmax + (SPLIT > 1 ? (wave / SPLIT) * 8 * SPLIT : 0);  // this task's exchange slots
    if (lmc_sym_nibbles(bins))
      quantize_task<G, NITER, DT, true, true, ROWS, 2, SPLIT>(a.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out,
                                                              nullptr, scale_out, sl, slice, xm);
    else
      quantize_task<G, NITER, DT, true, false, ROWS, 2, SPLIT>(a.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out,
                                                               nullptr, scale_out, sl, slice, xm);
  } else {
#pragma unroll 1
    for (int hq = 0; hq < 2; hq++) {
      const int q = 2 * oct + hq;
      quantize_task<G, NITER, DT, false, false>(a.src, p, tok0, Tc, q * 4, ovalid && q < a.TQ, false, a.C, maxf, nullptr,
                                                sym8_plane, scale_out + 4 * hq, sl);
    }
  }
}
