// k_fused.h -- the whole encode of a (chunk, plane) in ONE workgroup: quantise, then code (rows a5-a11).
//
// The two-kernel encode (k_quantize, then k_cdf_encode) runs an HBM-bound phase and a VALU-bound phase one after
// the other.  Here a workgroup of FUSED_WAVES waves owns one plane-chunk from the raw KV to the placed streams:
//   phase A  its waves quantise the plane-chunk's row octs (quantize_oct_fused, a wave = 8 token rows x all
//            channels: the row max never leaves the wave), symbols to the plane-chunk's workspace region, scales
//            to the blob;
//   barrier  (workgroup scope is enough: the region is written and read by waves of one CU)
//   phase B  wave w codes the group streams w, w + FUSED_WAVES, ... exactly as k_cdf_encode does
//            (encode_group_stream_counts: histogram, counts, table, interleaved rANS on the counts model -- this
//            kernel takes 256-token chunks only, lmc_api.hip hands a ragged last chunk to the two-kernel path) --
//            the symbols come back from L2;
//   placement: ONE look-back per plane-chunk over P granules per chunk, then every wave moves its own streams.
// Several workgroups share a CU (4.6 KiB of LDS per wave + 2 KiB of reciprocals, <= 64 VGPRs: 8 waves per SIMD) and are in different
// phases at any time, so the loads of one hide under the coding of the others.  Blobs are byte-identical to the
// two-kernel path (same device functions; tests/test_gpu_parity.py runs both).
//
// The look-back granules carry the launch's epoch (flag << 62 | epoch << 32 | value): a granule of another
// launch reads as "not published", so nothing has to zero them between jobs.
//
// Geometry: 256 < C <= 1024 channels per plane (G = 5..16 group streams, the 64-lane quantise tasks); other
// shapes take the two-kernel path (lmc_api.hip).
#pragma once
#include "k_encode_counts.h"
#include "k_quantize.h"

#define FUSED_MAX_G 16
#ifndef FUSED_WAVES
#define FUSED_WAVES 8  // waves per workgroup: 4 workgroups per CU
#endif
// Cache policy of the fused encode's streams.  The symbol workspace and the stream scratch are written and read back
// within a workgroup's life, and 1024 workgroups' worth of them (~300 MB) is about what L2 + Infinity Cache hold --
// so everything that is touched ONCE says so: the raw KV is loaded non-temporal (LMC_FUSED_NT_IN), the placed streams
// are stored non-temporal (LMC_PLACE_NT_OUT, k_encode.h), the coding pass reads its symbols for the last time
// non-temporal (LMC_SYM_NT, k_encode_counts.h).  Same box, alternating processes: 1.017-1.027 ms without, 0.982-0.988
// with the first, 0.968-0.975 with all three; a non-temporal re-read of the scratch slot on top changes nothing.
#ifndef LMC_FUSED_NT_IN
#define LMC_FUSED_NT_IN 1
#endif
#ifndef LMC_FUSED_PRIO_A
#define LMC_FUSED_PRIO_A 3  // wave priority while a wave fetches / quantises (phase A) ...
#endif
#ifndef LMC_FUSED_PRIO_B
#define LMC_FUSED_PRIO_B 0  // ... and while it codes (phase B)
#endif

struct FusedArgs {
  KvAddr src;
  EncodeArgs e;  // e.sym4 = workspace (written in phase A), e.agg = epoch-tagged plane granules [nchunks][P]
  u8* scale_base;
  long long scale_stride;
  u32 epoch;  // 1 .. 2^30 - 1
  // Head start (an experiment that did NOT pay; off by default, LMC_FUSED_PRE_STEP=n switches it on).  Doubling phase A
  // costs a whole k_quantize (+0.44 ms), as if the phases of different workgroups never overlapped; the hypothesis
  // was lock-step generations (every CU fetches, then every CU codes).  Here the plane-chunks of every `pre_step`-th
  // workgroup below `pre_limit` (the first generation) are quantised by a k_quantize launch in front of this kernel,
  // so that those workgroups start coding at once and the generations run out of phase.  Measured: 1.056 ms
  // without, 1.064-1.09 ms with pre_step 4 / 2 / 3 / 1 -- the phases were not in lock-step; what the fetch phase
  // costs is the wave slots its waves hold while they wait.
  u32 pre_limit, pre_step;
  // Staggered start (lmc_api.hip: 50 us, LMC_FUSED_STAGGER_US): every workgroup of a launch takes the same time, so
  // the four workgroups a CU holds tend to run their phases in lock-step for the whole launch -- all fetch (HBM busy,
  // VALU idle), then all code (VALU busy, HBM idle).  A first-generation workgroup (ticket < stagger_limit) draws its
  // rank among the workgroups of ITS CU (hardware id -> cu_rank[]) and holds its fetch back by rank x stagger_ticks
  // (s_memrealtime ticks, 10 ns): the CU's slots then run about a quarter period apart.  Measured over alternating
  // processes on one box: 1.046-1.12 ms without, 1.010-1.059 with 50 us (-3 % on average).  Keyed on the ticket
  // instead of the CU (rank = ticket / CUs) it does nothing: tickets follow dispatch order, which is not one
  // workgroup per CU at a time.
  u32 stagger_ticks, stagger_limit;
  u32* cu_rank;  // [4096] zero at launch: the workgroup with the launch's last ticket clears it again on its way out
};

__device__ __forceinline__ void aggE_store(unsigned long long* p, unsigned long long flag, u32 epoch, u32 v) {
  __hip_atomic_store(p, (flag << 62) | ((unsigned long long)epoch << 32) | (unsigned long long)v, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// lookback_exclusive (k_encode.h) over epoch-tagged granules.
__device__ __forceinline__ u32 lookback_exclusive_epoch(unsigned long long* agg, int idx0, u32 epoch, int lane, u32* status) {
  u32 excl = 0;
  if (idx0 > 0) {
    int base = idx0 - 1;
    u32 spins = 0;
    for (;;) {
      const int idx = base - lane;
      const unsigned long long v = idx >= 0 ? agg_load(agg + idx) : ((AGG_P << 62) | ((unsigned long long)epoch << 32));
      const bool ours = ((u32)(v >> 32) & 0x3fffffffu) == epoch;
      const u32 flag = ours ? (u32)(v >> 62) : (u32)AGG_X;
      const u64 mP = __ballot(flag == (u32)AGG_P), mX = __ballot(flag == (u32)AGG_X);
      const int first = mP ? __builtin_ctzll(mP) : 64;
      const u64 below = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
      if (mX & below) {
        if (++spins > (1u << 24)) {
          if (lane == 0) atomicOr(status, LMC_ST_LOOKBACK_TIMEOUT);
          break;
        }
        __builtin_amdgcn_s_sleep(10);  // ~0.3 us: a waiting wave should cost issue slots as rarely as possible
        continue;
      }
      excl += wave_sum_u32(lane <= first ? (u32)v : 0u);
      if (mP) break;
      base -= 64;
    }
  }
  return excl;
}

// One row oct (tokens t_first .. t_first + 7 of the plane) by one wave: quantize_task<64, NITER, DT, QUAD, NIB>
// (k_quantize.h) re-staged for 64 VGPRs -- the same arithmetic (quant_z2 / v_cvt_pk_u8_f32 on regular rows,
// quant_special on zero / inf / NaN rows), two rows in flight, one row quad of accumulators at a time: a byte
// plane stores each quad as soon as it is complete; a nibble plane parks the first quad's 8 * NITER dwords in
// the wave's idle LDS slice and merges them with the second (byte k = token k | token 4 + k << 4).
// FULL: every lane of every iteration holds channels of the plane (C == NITER * 512: Llama / Mistral GQA shapes), so
// no per-lane validity is tested and no register is zero-filled for absent channels.
template <int NITER, int DT, bool NIB, bool FULL = false>
__device__ __forceinline__ void quantize_oct_fused(const KvAddr& src, const u16* pbase, int tok0, int Tc, int t_first,
                                                   bool q1valid, int C, float maxf, u32* sym_out, u16* scale_out,
                                                   uint4* park, int lane) {
  long long coff[NITER];
  int c0[NITER];
  bool cval[NITER];
#pragma unroll
  for (int it = 0; it < NITER; it++) {
    c0[it] = (it * 64 + lane) * 8;
    cval[it] = FULL || c0[it] < C;
    const int h = c0[it] / src.D, d = c0[it] - h * src.D;
    coff[it] = (long long)h * src.stride_head + d;
  }
  const f32x2_t maxf2 = {maxf, maxf};
#pragma unroll
  for (int hq = 0; hq < 2; hq++) {
    u32 o[NITER][8];  // byte r = symbol of (token t_first + 4 hq + r, channel c0[it] + e); row 0 of the quad defines it
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += 2) {
      uint4 v[2][NITER];
      bool tv[2];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int t = t_first + 4 * hq + r0 + r;
        tv[r] = t < Tc;
#if LMC_EXP_TWICE & 16  // timing experiment: every row is the chunk's first row (L2 hits instead of HBM reads)
        const u16* rowp = pbase + (tv[r] ? lmc_tok_off(src, tok0 + (t & 7)) : 0);
#else
        const u16* rowp = pbase + (tv[r] ? lmc_tok_off(src, tok0 + t) : 0);
#endif
#pragma unroll
        for (int it = 0; it < NITER; it++) {
#if LMC_FUSED_NT_IN
          if (tv[r] && (FULL || cval[it])) v[r][it] = ld_global_u4_nt(rowp + coff[it]);
#else
          if (tv[r] && (FULL || cval[it])) v[r][it] = ld_global_u4(rowp + coff[it]);
#endif
          else v[r][it] = make_uint4(0, 0, 0, 0);
        }
      }
      u32 mrow[2];
#pragma unroll
      for (int r = 0; r < 2; r++) {
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
      wave_max2_u32(mrow[0], mrow[1]);  // wave-uniform from here on
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 2; r++)
          if (tv[r]) scale_out[4 * hq + r0 + r] = (u16)mrow[r];
      }
      float factor[2];
      bool special[2] = {false, false};
      bool slow = false;
      // both row maxes are wave-uniform: one scalar branch picks the short division for the pair -- and a max inside
      // its range is finite, non-zero and has a finite factor, so such a pair cannot be special: no tests at all
      const bool short_div = LMC_SHORT_ROW_DIV && row_div_in_range(mrow[0], DT) && row_div_in_range(mrow[1], DT);
      if (short_div) {
#pragma unroll
        for (int r = 0; r < 2; r++) factor[r] = row_div_short(maxf, h2f_rt(mrow[r], DT));
      } else {
        bool any_special = false;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const float sf = h2f_rt(mrow[r], DT);
          factor[r] = maxf / sf;  // IEEE fp32 division (lmc_device.h)
          special[r] = !(__builtin_fabsf(factor[r]) < __builtin_inff()) || !(sf < __builtin_inff());
          any_special |= special[r];
        }
        slow = __ballot(any_special) != 0;  // wave-uniform and rare
      }
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        if (!FULL && !cval[it]) continue;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
          if (!slow) {
            const f32x2_t f2 = {factor[r], factor[r]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const f32x2_t z = quant_z2(h_lo<DT>(w[k]), h_hi<DT>(w[k]), f2, maxf2);
              o[it][2 * k] = __builtin_amdgcn_cvt_pk_u8_f32(z.x, r0 + r, r0 + r ? o[it][2 * k] : 0u);
              o[it][2 * k + 1] = __builtin_amdgcn_cvt_pk_u8_f32(z.y, r0 + r, r0 + r ? o[it][2 * k + 1] : 0u);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float xl = h_lo<DT>(w[k]), xh = h_hi<DT>(w[k]);
              const u32 sl_ = (special[r] ? quant_special(xl, factor[r], maxf) : quant_fast(xl, factor[r], maxf)) & 0xffu;
              const u32 sh_ = (special[r] ? quant_special(xh, factor[r], maxf) : quant_fast(xh, factor[r], maxf)) & 0xffu;
              o[it][2 * k] = (r0 + r ? o[it][2 * k] : 0u) | sl_ << (8 * (r0 + r));
              o[it][2 * k + 1] = (r0 + r ? o[it][2 * k + 1] : 0u) | sh_ << (8 * (r0 + r));
            }
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      if (!FULL && !cval[it]) continue;
      if (NIB) {
        uint4* pk = park + (it * 2) * 64 + lane;
        if (hq == 0) {
          pk[0] = make_uint4(o[it][0], o[it][1], o[it][2], o[it][3]);
          pk[64] = make_uint4(o[it][4], o[it][5], o[it][6], o[it][7]);
        } else {
          const uint4 a0 = pk[0], a1 = pk[64];
          u32* dst = sym_out + c0[it];
          *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x | (o[it][0] << 4), a0.y | (o[it][1] << 4), a0.z | (o[it][2] << 4), a0.w | (o[it][3] << 4));
          *reinterpret_cast<uint4*>(dst + 4) = make_uint4(a1.x | (o[it][4] << 4), a1.y | (o[it][5] << 4), a1.z | (o[it][6] << 4), a1.w | (o[it][7] << 4));
        }
      } else {
        if (hq == 1 && !q1valid) continue;
        u32* dst = sym_out + (long long)hq * C + c0[it];  // the oct's row quads are adjacent [quad][channel] rows
        *reinterpret_cast<uint4*>(dst) = make_uint4(o[it][0], o[it][1], o[it][2], o[it][3]);
        *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[it][4], o[it][5], o[it][6], o[it][7]);
      }
    }
  }
}

template <int NITER, int DT, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_encode_fused(FusedArgs fa) {
  const EncodeArgs& a = fa.e;
  __shared__ __attribute__((aligned(16))) u32 lds_all[NW * (ENC_RING_DWORDS + CNT_TAB_DWORDS)];  // the staging rings, then the tables
  __shared__ __attribute__((aligned(16))) u32 rtab_lds[RTAB_DWORDS];  // reciprocals of the counts model's frequencies
  __shared__ u32 st_len[FUSED_MAX_G];  // exact byte length of the plane-chunk's group streams
  __shared__ u32 wg_excl;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // consecutive work items are the same plane of consecutive chunks (see k_cdf_encode): a plane-chunk's predecessors
  // in the look-back were taken at least nchunks workgroups earlier.  The item comes from a ticket, not from
  // blockIdx (EncodeArgs::ticket): a predecessor's workgroup has started, whatever order the hardware dispatches in.
  const u32 item = (u32)__builtin_amdgcn_readfirstlane((int)draw_ticket(a.ticket, a.ticket_base));
  const int chunk = (int)(item % (unsigned)a.nchunks), p = (int)(item / (unsigned)a.nchunks);
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);
  u32* const hist = lds_all + NW * ENC_RING_DWORDS + wave * CNT_TAB_DWORDS;  // this wave's table slice ...
  u16* const ring = reinterpret_cast<u16*>(lds_all + wave * ENC_RING_DWORDS);  // ... and staging ring

  rtab_to_lds(rtab_lds);  // visible to the coder waves behind the barrier that ends phase A
  if (fa.stagger_ticks && item < fa.stagger_limit) {
    __shared__ u32 cu_slot;
    if (threadIdx.x == 0) {
      const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_ID, XCC_ID
      const u32 key = ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);  // XCC | SE, SH, CU
      cu_slot = atomicAdd(&fa.cu_rank[key], 1u);
    }
    __syncthreads();
    // wave-uniform, in SGPRs: the wait loop is scalar code (32-bit tick arithmetic: a hold is < 2^32 x 10 ns)
    const u32 hold = (u32)__builtin_amdgcn_readfirstlane((int)(min(cu_slot, 3u) * fa.stagger_ticks));
    const u32 t0 = (u32)__builtin_amdgcn_s_memrealtime();
    while ((u32)__builtin_amdgcn_s_memrealtime() - t0 < hold) __builtin_amdgcn_s_sleep(32);
  }
  // ---- phase A: quantise the plane-chunk ----------------------------------------------------------------
  const bool head_start = item < fa.pre_limit && item % fa.pre_step == 0u;  // quantised by k_quantize already
  if (!head_start) {
    const int bins = (int)a.bins.b[p];
    const float maxf = (float)(bins / 2 - 1);
    const bool nib = lmc_sym_nibbles(bins);
    const bool full = a.C == NITER * 512;  // no absent channels: the variant without per-lane validity
    u32* const sym_pc = const_cast<u32*>(a.sym4) + ((long long)chunk * a.P + p) * a.sym_stride;
    u16* const scl = reinterpret_cast<u16*>(fa.scale_base + (long long)chunk * fa.scale_stride) + (long long)p * Tc;
    const int TO = (Tc + 7) >> 3;
    const u16* const pbase = lmc_plane_base(fa.src, p);
    uint4* const park = reinterpret_cast<uint4*>(hist);  // the wave's table slice is idle until phase B
    // Waves that fetch run at raised priority: their (few) instructions go first, so the loads are out early and
    // return under the other workgroups' coding.
    __builtin_amdgcn_s_setprio(LMC_FUSED_PRIO_A);
#if LMC_EXP_TWICE & 1
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++)
#endif
#pragma unroll 1
    for (int oct = wave; oct < TO; oct += NW) {
      const bool q1valid = 2 * oct + 1 < a.TQ;
      if (full) {  // wave-uniform
        if (nib)
          quantize_oct_fused<NITER, DT, true, true>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                    sym_pc + (long long)oct * a.C, scl + oct * 8, park, lane);
        else
          quantize_oct_fused<NITER, DT, false, true>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                     sym_pc + (long long)oct * 2 * a.C, scl + oct * 8, park, lane);
      } else if (nib)
        quantize_oct_fused<NITER, DT, true>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                            sym_pc + (long long)oct * a.C, scl + oct * 8, park, lane);
      else
        quantize_oct_fused<NITER, DT, false>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                             sym_pc + (long long)oct * 2 * a.C, scl + oct * 8, park, lane);
    }
    __builtin_amdgcn_s_setprio(LMC_FUSED_PRIO_B);
  }
  __syncthreads();  // symbols and scales of the plane-chunk are visible to the workgroup
#if LMC_EXP_TWICE & 128  // timing experiment: phase A only
  return;
#endif

  // ---- phase B: code this wave's group streams -----------------------------------------------------------
  const long long gid0 = ((long long)chunk * a.P + p) * a.G;
#pragma unroll 1
  for (int g = wave; g < a.G; g += NW) {
    PendingTile t;
    encode_group_stream_counts<LMC_COUNTS_LDSASM>(a, gid0 + g, hist, ring, rtab_lds, lane, t);
    if (lane == 0) st_len[g] = t.exact;
    wave_lds_fence();  // the next stream reuses this wave's LDS slices
  }
  __syncthreads();

  // ---- placement: one look-back per plane-chunk --------------------------------------------------------------
  u32 wg_total = 0;
  for (int g = 0; g < a.G; g++) wg_total += (st_len[g] + 15u) & ~15u;
  if (wave == 0) {
    unsigned long long* agg = a.agg + (long long)chunk * a.P;
    if (lane == 0 && p > 0) aggE_store(agg + p, AGG_A, fa.epoch, wg_total);
#if LMC_EXP_TWICE & 256  // timing experiment: no look-back (the streams land at offset 0 of their chunk: blobs are wrong)
    const u32 e = 0;
#else
    const u32 e = lookback_exclusive_epoch(agg, p, fa.epoch, lane, a.status);
#endif
    if (lane == 0) {
      aggE_store(agg + p, AGG_P, fa.epoch, e + wg_total);
      wg_excl = e;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int g = wave; g < a.G; g += NW) {
    u32 intra = 0;
    for (int k = 0; k < g; k++) intra += (st_len[k] + 15u) & ~15u;
    PendingTile t;
    t.chunk = chunk; t.pg = p * a.G + g; t.exact = st_len[g]; t.T = (u32)Tc;
#if LMC_EXP_TWICE & 64
    t.out = reinterpret_cast<const u16*>(a.scratch + ((gid0 + g) % 256) * (long long)a.cap);
#else
    t.out = reinterpret_cast<const u16*>(a.scratch + (gid0 + g) * (long long)a.cap);
#endif
#if !(LMC_EXP_TWICE & 8)
    place_stream<false>(a, t, wg_excl + intra, hist, lane);
#endif
  }
  // The chunk's last plane knows the chunk's size: header, static sections, size word (kept out of the stream loop:
  // inlined there, its loop invariants were hoisted over the loop and spilled by every wave).
  if (p == a.P - 1 && wave == 0) {
    const BlobOff bo = lmc_blob_off((u32)a.P, (u32)Tc, (u32)a.C, (u32)a.G, (u32)a.bins.rowpre[a.P]);
    u8* blob = a.blobs + (long long)chunk * a.blob_stride;
    write_blob_static(blob, bo, a, (u32)Tc, wg_excl + wg_total, lane);
    if (lane == 0) a.sizes[chunk] = bo.streams + wg_excl + wg_total;
  }
  if (fa.stagger_ticks && item == (u32)(a.nchunks * a.P) - 1u) {  // every first-generation rank was drawn long ago
    for (u32 i = threadIdx.x; i < 4096u; i += 64u * NW) fa.cu_rank[i] = 0u;
  }
}
