// k_fused.h -- the whole encode of a (chunk, plane) in ONE workgroup: quantise, then code (rows a5-a11).
//
// The two-kernel encode (k_quantize, then k_cdf_encode) runs an HBM-bound phase and a VALU-bound phase one after
// the other.  Here a workgroup of FUSED_WAVES waves owns one plane-chunk from the raw KV to the streams in the blob:
//   phase A   its waves quantise the plane-chunk's row octs (quantize_oct_fused, a wave = 8 token rows x all
//             channels: the row max never leaves the wave), symbols to the plane-chunk's workspace region, scales
//             to the blob;
//   barrier   (workgroup scope is enough: the region is written and read by waves of one CU)
//   pass 1    wave w takes the histograms of the group streams w, w + FUSED_WAVES (counts_hist_stream): the counts stay
//             in its registers, and every stream's ALLOCATION in the blob follows from them (lmc_format.h, v6: an
//             upper bound of its length);
//   placement ONE look-back per plane-chunk over P granules per chunk on the sum of the allocations: every stream of
//             the plane-chunk knows its final place BEFORE it is coded;
//   pass 2    head, table, interleaved rANS (counts_open_stream, counts_code_stream) -- the symbols come back from L2,
//             the words leave in 256-byte pieces for the blob itself.  No stream scratch, no second placement pass
//             (format v5 wrote every stream twice and read it once in between: 1.5 GB of fabric traffic per 16 k context).
// Several workgroups share a CU (4.6 KiB of LDS per wave + 2.5 KiB of reciprocals and bound table, <= 64 VGPRs: 8
// waves per SIMD) and are in different phases at any time, so the loads of one hide under the coding of the others.
// (Round 3 held a CU's first workgroups back by rank x 50 us so that its four slots would not run their phases in
// lock-step; with the look-back between the two passes the slots drift apart by themselves and the same hold-back
// costs 2 - 9 %: 0.925-0.933 ms without, 0.942-0.961 with, alternating rounds on one box.  Removed.)
// Blobs are byte-identical to the two-kernel path (same device functions; tests/test_gpu_parity.py runs both).
//
// The look-back granules carry the launch's epoch (flag << 62 | epoch << 32 | value): a granule of another
// launch reads as "not published", so nothing has to zero them between jobs.
//
// Work item = a run of whole planes of one chunk (FusedArgs::pl planes, ipc items per chunk), sized so that an item
// is about half a megabyte of raw KV whatever the plane width:
//   C <= 128   (GL = 16)  8 planes per item, <= 16 streams; a wave quantises four row octs at a time, 16 lanes each
//   C <= 256   (GL = 32)  4 planes per item, <= 16 streams; two row octs at a time
//   C <= 1024  (GL = 64)  1 plane, <= 16 streams (NITER = 1 or 2 channel runs per lane)
// Planes of more than 1024 channels (C = 4096 is BASELINE configs[0]) take the two-kernel path: a fused form for them
// (2 / 4 waves sharing a row oct, the counts of a wave's 4 / 8 streams parked in a global stash between the passes)
// was built in round 4, bit-exact, and never beat k_quantize + k_cdf_encode below eight generations of workgroups --
// which no BASELINE configuration reaches -- so round 5 removed it.  A ragged last chunk takes
// the two-kernel path as well.
#pragma once
#include "k_encode_counts.h"
#include "k_quantize.h"

#define FUSED_MAX_NS 16  // streams per work item
#ifndef FUSED_WAVES
#define FUSED_WAVES 8  // waves per workgroup: 4 workgroups per CU
#endif
// Cache policy.  The symbol workspace is written and read back within a workgroup's life, and 1024 workgroups' worth
// of it is what L2 + Infinity Cache should hold -- so everything that is touched ONCE says so: the raw KV is loaded
// non-temporal, the streams are stored non-temporal (counts_code_stream<true>), the coding pass reads its symbols for
// the last time non-temporal (LMC_SYM_LAST_LOAD, k_encode_counts.h).  Round 3, same box, alternating processes:
// 1.017-1.027 ms without, 0.968-0.975 with.
#ifndef LMC_FUSED_PRIO_A
#define LMC_FUSED_PRIO_A 3  // wave priority while a wave fetches / quantises (phase A): its few instructions go first
#endif

#ifdef LMC_EXP_TIMELINE  // experiments only: phase time stamps of every work item (tools/probes/fused_timeline.hip / .py)
__device__ unsigned long long g_fused_timeline[8192 * 8];
#define LMC_TL(k) do { if (threadIdx.x == 0 && tick < 8192u) g_fused_timeline[tick * 8u + (k)] = wall_clock64(); } while (0)
#else
#define LMC_TL(k) do { } while (0)
#endif
struct FusedArgs {
  KvAddr src;
  EncodeArgs e;  // e.sym4 = workspace (written in phase A), e.agg = epoch-tagged plane granules [nchunks][P]
  u8* scale_base;
  long long scale_stride;
  u32 epoch;  // 1 .. 2^30 - 1
  int pl, ipc;   // planes per work item, items per chunk = ceil(P / pl)
  u32 item_base; // the first work item of THIS launch: a job may be launched in ranges of items (plane ranges of all its
                 // chunks: lmc_store_pack_parts) -- same epoch, so a later range's look-back finds the earlier ranges'
                 // granules published
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

// The offset of token t's row.  PSRC (round 6): the kernel instance for a PAGED source whose block size is a power of
// two.  lmc_tok_off reads the token's slot with a vector load, and the wait for it is an s_waitcnt vmcnt(0) -- which on
// gfx9 also waits for the acknowledgement of every symbol store issued before: eight such stalls per row oct (the encode
// reading a paged cache ran 4.5 - 7 % behind the contiguous one).  t is wave-uniform, so the slot comes by a scalar load
// (lgkmcnt) and the row offset is scalar arithmetic.  The instances for every other source are compiled as before.
template <bool PSRC>
__device__ __forceinline__ long long fused_tok_off(const KvAddr& a, int t) {
  if constexpr (!PSRC) return lmc_tok_off(a, t);
  else {
    const unsigned long long ad = uniform_ptr64(a.slot_mapping + t);
    const u32 s = *(const __attribute__((address_space(4))) u32*)ad;  // (the low half: lmc_tok_off reads a slot as 32 bits too)
    const u32 bs = (u32)a.block_size;
    const u32 b = s >> (u32)__builtin_ctz(bs), w = s & (bs - 1u);
    return (long long)b * a.stride_block + (long long)w * a.stride_token;
  }
}

// One row oct (tokens t_first .. t_first + 7 of the plane) by one wave: quantize_task<64, NITER, DT, QUAD, NIB>
// (k_quantize.h) re-staged for 64 VGPRs -- the same arithmetic (quant_z2 / v_cvt_pk_u8_f32 on regular rows,
// quant_special on zero / inf / NaN rows), two rows in flight, one row quad of accumulators at a time: a byte
// plane stores each quad as soon as it is complete; a nibble plane parks the first quad's 8 * NITER dwords in
// the wave's idle LDS slice and merges them with the second (byte k = token k | token 4 + k << 4).
// FULL: every lane of every iteration holds channels of the plane (C == NITER * 512: Llama / Mistral GQA
// shapes), so no per-lane validity is tested and no register is zero-filled for absent channels.
// ALLROWS: all eight rows of the oct are tokens of the chunk (every oct but the last one of a chunk whose length is no
// multiple of 8): no per-row test, no zero fill -- what the kernel had when it only took 256-token chunks.
template <int NITER, int DT, bool NIB, bool FULL = false, bool ALLROWS = true, bool PSRC = false>
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
        tv[r] = ALLROWS || t < Tc;
        const u16* rowp = pbase + (tv[r] ? fused_tok_off<PSRC>(src, tok0 + t) : 0);
#pragma unroll
        for (int it = 0; it < NITER; it++) {
          if (tv[r] && (FULL || cval[it])) v[r][it] = ld_global_u4_nt(rowp + coff[it]);  // streamed once
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
      u32 slow = 0;  // (an integer, not a bool: a bool merged over the branch below becomes a lane mask, and testing it
                     // costs two vector instructions in front of every block of eight elements)
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
        slow = __ballot(any_special) != 0 ? 1u : 0u;  // wave-uniform and rare
      }
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        if (!FULL && !cval[it]) continue;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
          // (the test stays inside the loops over runs and rows -- a wave-uniform test and two taken scalar branches per run
          // and row, the regular path jumping over the special-row code.  ONE branch around the whole pair's conversions
          // was built in round 5: the scheduler then interleaves both runs and both rows of the straight-line regular path
          // and the kernel spills 184 bytes per lane instead of 20.)
          if (slow == 0u) {
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

// ---- the plane's histogram taken WHILE it is quantised (round 6) ------------------------------------------------------
// Pass 1 used to read every symbol back from the workspace only to count it: one of the symbols' two re-reads
// (0.64 GB of the fused kernel's 4.6 GB of fabric traffic per 16 k context).  For planes of 257 .. 1024 channels
// (GL == 64: one plane per work item) the quantising waves now count the symbols they hold in registers into ONE
// histogram of the whole plane, which lives in the workgroup's eight 4 KiB table slices (idle until pass 2):
//   <= 16 symbols  u16 counters; slice b = it * 4 + e / 2 is [16 symbols][64 lanes] dwords, lane = the QUANTISING lane
//                  (channel (it * 64 + lane) * 8 + e), half e & 1: a wave's 64 lanes hit 64 banks, and a nibble that
//                  sits at bits 8 .. 11 of its dword is a row address by one v_and_or_b32 (the slices are 4 KiB aligned)
//   more symbols   u8 counters; block b = it * 2 + e / 4 is [32 symbols][64 lanes] dwords, byte e & 3.  Token 0 of the
//                  chunk is left out, so a counter holds at most T - 1 <= 255 and never carries into its neighbour;
//                  pass 1 adds the token back (counts_from_plane_hist).
// The counters are LDS atomics (eight waves add into the same plane); pass 1 then reads 16 / 32 counters per lane instead
// of 32 / 64 workspace dwords and 256 ds_add of its own.  The instruction count is the same (2 VALU + 1 ds_add per
// symbol, moved from pass 1 into phase A); the symbols are read once, by pass 2.
// Nibble planes pair rows (q, q + 4) -- the two tokens that share a byte of the workspace dword -- instead of parking
// the first row quad in the table slice: the slices are the histogram now.  The high row costs one v_lshl_or_b32 per
// element more than the byte insert of v_cvt_pk_u8_f32 alone.
#ifndef LMC_FUSED_HIST_A
#define LMC_FUSED_HIST_A 1
#endif
#ifndef LMC_FUSED_HIST_A_BYTE
#define LMC_FUSED_HIST_A_BYTE 1  // ... for the planes with more than 16 symbols as well
#endif
#ifndef LMC_FUSED_HIST_A_NARROW
#define LMC_FUSED_HIST_A_NARROW 1  // ... and for items of several narrow planes (C <= 256)
#endif
// quantize_oct_fused with the histogram: one row oct by one wave, symbols to the workspace AND into the plane's counters.
// skip0: the oct holds token 0 of the chunk (byte planes leave it out of the counters).
template <int NITER, int DT, bool NIB, bool FULL, bool ALLROWS, bool PSRC = false>
__device__ __forceinline__ void quantize_oct_hist(const KvAddr& src, const u16* pbase, int tok0, int Tc, int t_first,
                                                  bool q1valid, bool skip0, int C, float maxf, u32* sym_out, u16* scale_out,
                                                  int lane) {
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
  const u32 lane4 = 4u * (u32)lane;
  u32 o[NITER][8];  // NIB: the oct's workspace dwords (byte q = token q | token q + 4 << 4); else one row quad's (byte r = token 4 hq + r)
  u32 ad[2] = {lane4, lane4};  // two address registers in turn: a symbol's SDWA write does not wait for the previous ds_add's read
  auto hist_dword = [&](auto it_tag, auto e_tag, auto hq_tag, u32 od) {
    constexpr int it = decltype(it_tag)::value, e = decltype(e_tag)::value, hq = decltype(hq_tag)::value;
    // (wave-uniform tests: the rows of a partial last oct, token 0 of a byte plane)
    plane_hist_dword<NIB, it, e, hq>(od, ad, [&](int row) { return (ALLROWS || t_first + row < Tc) && (NIB || !(row == 0 && skip0)); });
  };
#pragma unroll
  for (int q = 0; q < 4; q++) {
    // the pair's rows: NIB (q, q + 4); else (2q, 2q + 1), i.e. rows r0, r0 + 1 of quad hq
    const int rowi[2] = {NIB ? q : 2 * q, NIB ? q + 4 : 2 * q + 1};
    uint4 v[2][NITER];
    bool tv[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int t = t_first + rowi[r];
      tv[r] = ALLROWS || t < Tc;
      const u16* rowp = pbase + (tv[r] ? fused_tok_off<PSRC>(src, tok0 + t) : 0);
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        if (tv[r] && (FULL || cval[it])) v[r][it] = ld_global_u4_nt(rowp + coff[it]);  // streamed once
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
        if (tv[r]) scale_out[rowi[r]] = (u16)mrow[r];
    }
    float factor[2];
    bool special[2] = {false, false};
    u32 slow = 0;  // (an integer: see quantize_oct_fused)
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
      slow = __ballot(any_special) != 0 ? 1u : 0u;  // wave-uniform and rare
    }
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      if (!FULL && !cval[it]) continue;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
        // byte of the element's dword, and whether this row is the first to write the dword / its high nibble
        const int bpos = NIB ? q : (2 * q + r) & 3;
        const bool fresh = NIB ? (q == 0 && r == 0) : bpos == 0;
        const bool high = NIB && r == 1;
        if (slow == 0u) {
          const f32x2_t f2 = {factor[r], factor[r]};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const f32x2_t z = quant_z2(h_lo<DT>(w[k]), h_hi<DT>(w[k]), f2, maxf2);
            if (high) {
              // (opaque: left to itself the compiler re-associates the four high rows of a dword into an accumulator of
              // their own -- 16 more live registers, spilled -- and merges at the end)
              asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(o[it][2 * k]) : "v"(__builtin_amdgcn_cvt_pk_u8_f32(z.x, 0, 0u)), "n"(8 * bpos + 4));
              asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(o[it][2 * k + 1]) : "v"(__builtin_amdgcn_cvt_pk_u8_f32(z.y, 0, 0u)), "n"(8 * bpos + 4));
            } else {
              o[it][2 * k] = __builtin_amdgcn_cvt_pk_u8_f32(z.x, bpos, fresh ? 0u : o[it][2 * k]);
              o[it][2 * k + 1] = __builtin_amdgcn_cvt_pk_u8_f32(z.y, bpos, fresh ? 0u : o[it][2 * k + 1]);
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float xl = h_lo<DT>(w[k]), xh = h_hi<DT>(w[k]);
            const u32 sl_ = (special[r] ? quant_special(xl, factor[r], maxf) : quant_fast(xl, factor[r], maxf)) & 0xffu;
            const u32 sh_ = (special[r] ? quant_special(xh, factor[r], maxf) : quant_fast(xh, factor[r], maxf)) & 0xffu;
            o[it][2 * k] = (fresh ? 0u : o[it][2 * k]) | sl_ << (8 * bpos + (high ? 4 : 0));
            o[it][2 * k + 1] = (fresh ? 0u : o[it][2 * k + 1]) | sh_ << (8 * bpos + (high ? 4 : 0));
          }
        }
      }
    }
    if (NIB ? q == 3 : (q & 1) == 1) {  // the oct's (NIB) / the quad's dwords are complete: to the workspace, into the counters
      const int hq = NIB ? 0 : q >> 1;
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        if (!FULL && !cval[it]) continue;
        if (!NIB && hq == 1 && !q1valid) continue;
        u32* dst = sym_out + (NIB ? 0ll : (long long)hq * C) + c0[it];  // (byte planes: the oct's row quads are adjacent [quad][channel] rows)
        *reinterpret_cast<uint4*>(dst) = make_uint4(o[it][0], o[it][1], o[it][2], o[it][3]);
        *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[it][4], o[it][5], o[it][6], o[it][7]);
      }
#ifndef LMC_EXP_SKIP_HIST  // (instruction-count experiments only: the blobs are garbage without the counters)
      static_for<NITER>([&](auto it_tag) {
        constexpr int it = decltype(it_tag)::value;
        if (FULL || cval[it]) {
          static_for<8>([&](auto e_tag) {
            constexpr int e = decltype(e_tag)::value;
            if (q < 2) hist_dword(it_tag, e_tag, IntTag<0>{}, o[it][e]);
            else hist_dword(it_tag, e_tag, IntTag<1>{}, o[it][e]);
          });
        }
      });
#endif
    }
  }
}

template <int GL, int NITER, int DT, int NW, bool PSRC = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_encode_fused(FusedArgs fa) {
  static_assert(GL == 64 || NITER == 1, "narrow planes: one channel run per lane");
  const EncodeArgs& a = fa.e;
  // (4 KiB aligned, the 4 KiB table slices first: every slice starts at a multiple of 4 KiB, which the row addressing
  // of the counts coder uses -- row_addr_cnt ALIGNED)
  // (GL == 64: the eight slices together are the plane's histogram during phase A and pass 1 -- PLANE_HIST_DWORDS, at LDS
  // address 0: plane_hist_row)
  constexpr bool HISTA = LMC_FUSED_HIST_A && GL == 64 && NW * CNT_TAB_DWORDS == PLANE_HIST_DWORDS;
  // ... and narrow planes (GL < 64: several planes per item): the same 32 KiB hold the counters of ALL the item's planes,
  // plane pj's lane group on the virtual quantising lanes pj * GL .. (k_hist.h) -- when the item's planes fit 128 of them
  // GL == 32 only (planes of 129 .. 256 channels): a wave pass quantises 64 / GL row octs of one plane, whose lane groups add
  // into the SAME columns -- the same LDS bank.  Two groups (GL = 32) cost a 2-way conflict and the histogram still wins
  // 5 % (0.251 - 0.256 vs 0.262 - 0.270 ms, 32 layers x 2 heads x 128, 16 k tokens, five alternations); four groups
  // (GL = 16) make every ds_add a 4-way conflict: level at C = 128, + 18 % at C = 64 (profiles/r06_experiments.md).
  constexpr bool HISTN = LMC_FUSED_HIST_A && LMC_FUSED_HIST_A_NARROW && GL == 32 && NW * CNT_TAB_DWORDS == PLANE_HIST_DWORDS;
  __shared__ __attribute__((aligned(HISTA ? 32768 : 4096))) u32 lds_all[NW * (CNT_TAB_DWORDS + CNT_RING_DWORDS)];  // the tables, then the staging buffers
  __shared__ __attribute__((aligned(16))) u32 rtab_lds[RTAB_LDS_DWORDS];  // reciprocals of the counts model's frequencies, bound table
  __shared__ u32 st_alloc[FUSED_MAX_NS];  // allocation of the item's group streams
  __shared__ u32 wg_excl;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // consecutive work items are the same planes of consecutive chunks (see k_cdf_encode): an item's predecessors
  // in the look-back were taken at least nchunks workgroups earlier.  The item comes from a ticket, not from
  // blockIdx (EncodeArgs::ticket): a predecessor's workgroup has started, whatever order the hardware dispatches in.
  const u32 tick = (u32)__builtin_amdgcn_readfirstlane((int)draw_ticket(a.ticket, a.ticket_base));
  const u32 item = fa.item_base + tick;
  LMC_TL(0);
#ifdef LMC_EXP_TIMELINE
  if (threadIdx.x == 0 && tick < 8192u) {
    u32 hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_fused_timeline[tick * 8u + 6u] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
  const int chunk = (int)(item % (unsigned)a.nchunks), it = (int)(item / (unsigned)a.nchunks);
  const int p0 = it * fa.pl, np = min(fa.pl, a.P - p0);  // the item's planes
  const int NS = np * a.G;                               // ... and streams: j -> plane p0 + j / G, group j % G
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = a.chunk_tokens;  // lmc_api.hip hands this kernel the job's FULL chunks only: 32 .. 256 tokens each (round 5; 256 before)
  u32* const hist = lds_all + wave * CNT_TAB_DWORDS;  // this wave's table slice ...
  u16* const ring = reinterpret_cast<u16*>(lds_all + NW * CNT_TAB_DWORDS + wave * CNT_RING_DWORDS);  // ... and staging buffer

  rtab_to_lds(rtab_lds);  // visible to the coder waves behind the barrier that ends phase A
  // the item's plane takes its histogram while it is quantised (wave-uniform per item: GL == 64 items are one plane)
  bool hist_a = false;
  if constexpr (HISTN) {
    typedef __attribute__((address_space(3))) u32* lds_u32w;
    if ((u32)(size_t)(lds_u32w)lds_all != 0u) __builtin_trap();
    hist_a = np * GL <= 128;  // (wave-uniform; false only with an LMC_FUSED_PL override that packs more planes into an item)
    if (hist_a) {
      uint4* z = reinterpret_cast<uint4*>(lds_all);
#pragma unroll
      for (int i = 0; i < PLANE_HIST_DWORDS / 4 / (64 * NW); i++) z[i * 64 * NW + threadIdx.x] = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
  }
  if constexpr (HISTA) {
    typedef __attribute__((address_space(3))) u32* lds_u32w;
    if ((u32)(size_t)(lds_u32w)lds_all != 0u) __builtin_trap();  // (static layout: plane_hist_row builds addresses from 0)
    const int bins0 = (int)a.bins.b[p0];
    hist_a = LMC_FUSED_HIST_A_BYTE || lmc_sym_nibbles(bins0);  // (a constant with LMC_FUSED_HIST_A_BYTE: quantize_oct_fused is not instantiated then)
    if (hist_a) {
      uint4* z = reinterpret_cast<uint4*>(lds_all);
#pragma unroll
      for (int i = 0; i < PLANE_HIST_DWORDS / 4 / (64 * NW); i++) z[i * 64 * NW + threadIdx.x] = make_uint4(0, 0, 0, 0);
      __syncthreads();  // (every wave adds everywhere)
    }
  }
  // A size word of 0 says "this chunk's encode did not finish" to whoever reads the words next (k_offload, k_pack_scan,
  // the host).  The workgroup that will write the chunk's size clears the word of whatever job used it before -- the
  // SAME workgroup, so that both stores pass through one XCD's L2 in program order (clearing from the chunk's first
  // item was tried: the two workgroups may sit on different XCDs, whose L2s write back in either order, and a test
  // caught a size of 0).  No memset dispatch in front of every job.
  if (p0 + np == a.P && threadIdx.x == 64 * (NW - 1)) a.sizes[chunk] = 0u;
  // ---- phase A: quantise the item's planes ------------------------------------------------------------------
  {
    const int TO = (Tc + 7) >> 3;  // row octs of a plane-chunk
    uint4* const park = reinterpret_cast<uint4*>(hist);  // the wave's table slice is idle until pass 1
    u8* const scl0 = fa.scale_base + (long long)chunk * fa.scale_stride;
    // Waves that fetch run at raised priority: their (few) instructions go first, so the loads are out early and
    // return under the other workgroups' coding.
    __builtin_amdgcn_s_setprio(LMC_FUSED_PRIO_A);
    if constexpr (GL < 64) {
      // narrow planes: a wave takes RPW row octs of one plane at a time, GL lanes each (quantize_task, k_quantize.h)
      constexpr int RPW = 64 / GL;                // octs per wave pass
      const int OG = (TO + RPW - 1) / RPW;        // oct groups per plane (the last one may be partial: Tc / 8 need not divide)
      const int sub = lane / GL, sl = lane % GL;
#pragma unroll 1
      for (int task = wave; task < np * OG; task += NW) {
        const int p = p0 + task / OG;
        int oct = (task % OG) * RPW + sub;
        const bool ovalid = oct < TO;             // (per lane group)
        if (!ovalid) oct = 0;
        const int bins = (int)a.bins.b[p];
        const float maxf = (float)(bins / 2 - 1);
        const bool nib = lmc_sym_nibbles(bins);
        u32* const sym_out = const_cast<u32*>(a.sym4) + ((long long)chunk * a.P + p) * a.sym_stride + (long long)oct * (nib ? 1 : 2) * a.C;
        u16* const scale_out = reinterpret_cast<u16*>(scl0) + (long long)p * Tc + oct * 8;
        const bool q1valid = 2 * oct + 1 < a.TQ;
        if (HISTN && hist_a) {
          const u32 vq = (u32)((task / OG) * GL + sl);  // the lane's virtual quantising lane within the item (k_hist.h)
          const u32 hl4 = 4u * (vq & 63u);
          const int hit = __builtin_amdgcn_readfirstlane((int)(((u32)(task / OG) * GL) >> 6));  // (GL lanes never straddle 64)
          if (nib) quantize_task<GL, 1, DT, true, true, 4, 2, 1, true>(fa.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out, nullptr, scale_out, sl, 0, nullptr, hl4, hit);
          else quantize_task<GL, 1, DT, true, false, 4, 2, 1, true>(fa.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out, nullptr, scale_out, sl, 0, nullptr, hl4, hit);
        } else if (nib) quantize_task<GL, 1, DT, true, true, 4, 2>(fa.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out, nullptr, scale_out, sl);
        else quantize_task<GL, 1, DT, true, false, 4, 2>(fa.src, p, tok0, Tc, oct * 8, ovalid, q1valid, a.C, maxf, sym_out, nullptr, scale_out, sl);
      }
    } else {
      const int p = p0;
      const int bins = (int)a.bins.b[p];
      const float maxf = (float)(bins / 2 - 1);
      const bool nib = lmc_sym_nibbles(bins);
      const bool full = a.C == NITER * 512;  // no absent channels: the variant without per-lane validity
      u32* const sym_pc = const_cast<u32*>(a.sym4) + ((long long)chunk * a.P + p) * a.sym_stride;
      u16* const scl = reinterpret_cast<u16*>(scl0) + (long long)p * Tc;
      const u16* const pbase = lmc_plane_base(fa.src, p);
#ifdef LMC_EXP_SKIP_PHASE_A
      if (false)
#endif
#pragma unroll 1
      for (int oct = wave; oct < TO; oct += NW) {
        const bool q1valid = 2 * oct + 1 < a.TQ;
        const bool allrows = oct * 8 + 8 <= Tc;  // wave-uniform: false only for the last oct of a chunk of 8 k + r tokens
        if (HISTA && (LMC_FUSED_HIST_A_BYTE || hist_a)) {
          // round 6: symbols to the workspace AND into the plane's counters (quantize_oct_hist)
          u32* const so = sym_pc + (long long)oct * (nib ? 1 : 2) * a.C;
          if (!allrows) {
            if (nib) quantize_oct_hist<NITER, DT, true, false, false, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
            else quantize_oct_hist<NITER, DT, false, false, false, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
          } else if (full) {
            if (nib) quantize_oct_hist<NITER, DT, true, true, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
            else quantize_oct_hist<NITER, DT, false, true, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
          } else if (nib) quantize_oct_hist<NITER, DT, true, false, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
          else quantize_oct_hist<NITER, DT, false, false, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, oct == 0, a.C, maxf, so, scl + oct * 8, lane);
        } else if constexpr (HISTA && LMC_FUSED_HIST_A_BYTE) {
          // (unreachable: every plane of a GL == 64 item takes the branch above)
        } else if (!allrows) {  // (the variant with per-lane channel validity takes the per-row test as well: two instances, not four)
          if (nib)
            quantize_oct_fused<NITER, DT, true, false, false, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                              sym_pc + (long long)oct * a.C, scl + oct * 8, park, lane);
          else
            quantize_oct_fused<NITER, DT, false, false, false, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                               sym_pc + (long long)oct * 2 * a.C, scl + oct * 8, park, lane);
        } else if (full) {  // wave-uniform
          if (nib)
            quantize_oct_fused<NITER, DT, true, true, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                      sym_pc + (long long)oct * a.C, scl + oct * 8, park, lane);
          else
            quantize_oct_fused<NITER, DT, false, true, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                       sym_pc + (long long)oct * 2 * a.C, scl + oct * 8, park, lane);
        } else if (nib)
          quantize_oct_fused<NITER, DT, true, false, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                     sym_pc + (long long)oct * a.C, scl + oct * 8, park, lane);
        else
          quantize_oct_fused<NITER, DT, false, false, true, PSRC>(fa.src, pbase, tok0, Tc, oct * 8, q1valid, a.C, maxf,
                                                      sym_pc + (long long)oct * 2 * a.C, scl + oct * 8, park, lane);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();  // symbols and scales of the item are visible to the workgroup
  LMC_TL(1);

  // ---- pass 1: the counts of this wave's group streams, and from them the streams' allocations ----------------
  // stream j of the item = group j % G of plane p0 + j / G
  auto stream_of = [&](int j) -> CountsStream {
    const int pj = (int)((u32)j / (u32)a.G);
    return counts_stream_of(a, chunk, p0 + pj, j - pj * a.G, lane);
  };
  CountsState cs0, cs1;  // of stream `wave`, and of stream `wave + NW`
  auto pass1 = [&](int j, CountsState& cs) {
    if (j < NS) {  // wave-uniform
      const CountsStream s = stream_of(j);
      u32 alloc;
      // (the counters are there; vq_base = the plane's first virtual quantising lane within the item)
      if (HISTA && (LMC_FUSED_HIST_A_BYTE || hist_a)) alloc = counts_hist_stream<true, true>(a, s, hist, rtab_lds + RTAB_DWORDS, lane, cs);
      else if (HISTN && hist_a) alloc = counts_hist_stream<true, true>(a, s, hist, rtab_lds + RTAB_DWORDS, lane, cs, (u32)(s.p - p0) * GL);
      else alloc = counts_hist_stream<true>(a, s, hist, rtab_lds + RTAB_DWORDS, lane, cs);
      if (lane == 0) st_alloc[j] = alloc;
    }
  };
  pass1(wave, cs0);
  pass1(wave + NW, cs1);
  __syncthreads();
  LMC_TL(2);

  // ---- placement: one look-back per item, BEFORE the streams are coded ------------------------------------------
  u32 wg_total = 0;
  for (int j = 0; j < NS; j++) wg_total += st_alloc[j];
  if (wave == 0) {
    unsigned long long* agg = a.agg + (long long)chunk * fa.ipc;
    if (lane == 0 && it > 0) aggE_store(agg + it, AGG_A, fa.epoch, wg_total);
    const u32 e = lookback_exclusive_epoch(agg, it, fa.epoch, lane, a.status);
    if (lane == 0) {
      aggE_store(agg + it, AGG_P, fa.epoch, e + wg_total);
      wg_excl = e;
    }
  }
  __syncthreads();
  LMC_TL(3);

  // ---- pass 2: every stream is coded at its final place -----------------------------------------------------------
  const BlobOff bo = lmc_blob_off((u32)a.P, (u32)Tc, (u32)a.G);
  u8* const blob = a.blobs + (long long)chunk * a.blob_stride;
  auto pass2 = [&](int j, const CountsState& cs) {
    if (j < NS) {
      u32 beg = wg_excl;
      for (int k = 0; k < j; k++) beg += st_alloc[k];
      const u32 alloc = st_alloc[j];
      const CountsStream s = stream_of(j);
      u8* const out = blob + bo.streams + beg;
      counts_open_stream(s, cs, out, hist, lane);
      const u32 exact = cs.head + counts_code_stream<true>(a, s, hist, ring, rtab_lds, lane, reinterpret_cast<u16*>(out + cs.head));
      const u32 padded = (exact + 15u) & ~15u;
      if (alloc > padded) zero_fill16(out + padded, alloc - padded, lane);
      if (lane == 0) {
        u32* d = reinterpret_cast<u32*>(blob + bo.gdir) + 2 * (s.p * a.G + s.g);
        d[0] = beg;
        d[1] = beg + exact;
        if (exact > alloc) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);  // the bound is a theorem: never
      }
      wave_lds_fence();  // the next stream reuses this wave's LDS slices
    }
  };
  pass2(wave, cs0);
  pass2(wave + NW, cs1);
#ifdef LMC_EXP_TIMELINE
  LMC_TL(4);   // wave 0 is through
  __syncthreads();
  LMC_TL(5);   // ... every wave is
#endif
  // The chunk's last item knows the chunk's size: header, static sections, size word.
  if (p0 + np == a.P && wave == NW - 1) {
    write_blob_static(blob, bo, a, (u32)Tc, wg_excl + wg_total, lane);
    if (lane == 0) a.sizes[chunk] = bo.streams + wg_excl + wg_total;
  }
}
