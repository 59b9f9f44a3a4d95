// k_encode_counts.h -- the group-stream coder under LMC_MODEL_COUNTS (include/lmc_format.h): chunks of exactly
// 256 tokens, the reference's chunk size (lmcache/config.py: chunk_size 256).  Stands where
// torchac_cuda.calculate_cdf + encode_fast_new stand (cachegen_encoder.py:241-262, 287-289); the counts it
// stores are what the reference's CDF is a function of (lmc_calculate_cdf still exports that CDF).
//
// Why a second model.  Under CDF16 the token step divides the 32-bit state by a 16-bit frequency: two
// conversions, v_rcp_f32, a biased fma, a truncation and a compare-and-carry correction (rans_put) -- 32 of the
// step's ~72 VALU issue cycles, next to 14 for the word append.  A 256-token channel's counts already sum to
// 2^8: with freq = 2 * count out of 2^9 the frequency is a function of `count` alone, so
//   * its reciprocal is ONE entry of a 257-entry table shared by the workgroup (2 KiB of LDS): the quotient is
//     v_mul_hi_u32 + v_lshrrev_b32, exact for every state below 2^31 (lmc_rans_magic; the factor 2 keeps
//     freq >= 2, which is what lets a 32-bit multiplier do it) -- fetched a token ahead, off the state's chain;
//   * the per-(symbol, lane) entry is one dword (<= 16 symbols: count << 23 | start) or one u16 (start << 8 | count):
//     ONE LDS read per token instead of two;
//   * the code length is the channel's empirical entropy (0.1-0.2 % below CDF16 on rand / randn data).
// Token step of a <= 16-symbol plane: 11 VALU (v_cmpx_sdwa, mbcnt x 2, lshl_add, lshr | mul_hi, lshr, lshr, mad,
// add_sdwa | the next entry's reciprocal address) + 2-3 for the row address of the entry after that.
#pragma once
#include "k_encode.h"

// ---- reciprocals of the frequencies 2 * count, count = 0 .. 256 ---------------------------------------------
// entry = {magic, (512 - 2 * count) | shift << 24}: x / (2 count) = mulhi(x, magic) >> shift (the shift is fetched
// from the entry's top byte by an SDWA operand select), and 512 - freq is the multiplier of the state update
// x += q * (512 - freq) (v_mad_u32_u24 reads the low 24 bits of its operands: no extraction either).
#define RTAB_ENTRIES 257
#define RTAB_DWORDS (2 * RTAB_ENTRIES + 2)  // 2064 B: a multiple of 16
struct RansRtab {
  u32 v[RTAB_DWORDS];
};
constexpr RansRtab make_rans_rtab() {
  RansRtab t{};
  for (u32 c = 1; c <= 256u; c++) {
    const u32 f = 2u * c;
    u32 l = 1;
    while ((1u << l) < f) l++;
    const u64 num = 1ull << (31u + l);
    t.v[2 * c] = (u32)((num + f - 1u) / f);   // lmc_rans_magic
    t.v[2 * c + 1] = (512u - f) | ((l - 1u) << 24);
  }
  return t;
}
__device__ const RansRtab g_rans_rtab = make_rans_rtab();

// every thread of the workgroup takes part; the caller synchronises before the first coder step
__device__ __forceinline__ void rtab_to_lds(u32* rtab_lds) {
  for (u32 i = threadIdx.x; i < RTAB_DWORDS; i += blockDim.x) rtab_lds[i] = g_rans_rtab.v[i];
}

// how far ahead the token loop requests its table reads (encode_group_stream_counts): 0: one token, in front of the
// ring store; 1: entry two tokens / reciprocal one token ahead, behind the ring store; 2: four / two tokens ahead
#ifndef LMC_COUNTS_LDSASM
#define LMC_COUNTS_LDSASM 2
#endif

// timing experiments (tools/probes): bit 0 the quantise phase twice, bit 1 the histogram pass twice, bit 2 the
// coding pass twice (all three leave the blobs unchanged), bit 3 no placement copy (blobs incomplete); bits 4-6 put
// the raw rows / the symbol workspace / the stream scratch of the fused kernel on a few aliased regions that stay
// in L2 (what the kernel costs without that HBM traffic; blobs are garbage)
// the coding pass reads a symbol dword for the last time: with LMC_SYM_NT the load says so (non-temporal)
#ifndef LMC_SYM_NT
#define LMC_SYM_NT 1
#endif
#if LMC_SYM_NT
#define LMC_SYM_LAST_LOAD(p) __builtin_nontemporal_load((const LMC_GLOBAL u32*)(p))
#else
#define LMC_SYM_LAST_LOAD(p) (*(p))
#endif
#ifndef LMC_EXP_TWICE
#define LMC_EXP_TWICE 0
#endif

#define CNT_TAB_DWORDS 1024  // per wave: counters, then (aliased) the table: [16][64] u32, or [32][64] u16

// LDS byte address base + (field << SHIFT), field = WIDTH bits of w at bit POS, in TWO instructions whatever the
// position (the coders are bound by the NUMBER of VALU instructions: SQ_ACTIVE_INST_VALU is 4.0 SIMD cycles per
// instruction of any class, profiles/r03_*_pmc.md).  Opaque asm: left to itself the compiler canonicalises the
// middle fields to shift + and + add.
template <int POS, int WIDTH, int SHIFT>
__device__ __forceinline__ u32 row_addr_sh(u32 w, u32 base) {
  constexpr u32 FIELD = ((1u << WIDTH) - 1u);
  u32 r;
  if constexpr (POS == SHIFT) {
    asm("v_and_b32_e32 %0, %2, %1\n\tv_add_u32_e32 %0, %0, %3" : "=&v"(r) : "v"(w), "s"(FIELD << SHIFT), "v"(base));
  } else if constexpr (POS + WIDTH == 32) {
    asm("v_lshrrev_b32_e32 %0, %2, %1\n\tv_lshl_add_u32 %0, %0, %3, %4"
        : "=&v"(r) : "v"(w), "n"(POS), "n"(SHIFT), "v"(base));
  } else {
    asm("v_bfe_u32 %0, %1, %2, %3\n\tv_lshl_add_u32 %0, %0, %4, %5"
        : "=&v"(r) : "v"(w), "n"(POS), "n"(WIDTH), "n"(SHIFT), "v"(base));
  }
  return r;
}
// ... of token I (0..31) of a 32-token block of workspace dwords (k_quantize.h formats).  Nibble planes: rows of
// 256 B (dword entries), byte planes: rows of 128 B (u16 entries).
template <bool NIB, int I>
__device__ __forceinline__ u32 row_addr_cnt(const u32* w, u32 base) {
  if constexpr (NIB) return row_addr_sh<8 * (I & 3) + 4 * ((I >> 2) & 1), 4, 8>(w[I >> 3], base);
  else return row_addr_sh<8 * (I & 3), 8, 7>(w[I >> 2], base);
}

typedef u32 u32x2_t __attribute__((ext_vector_type(2)));

// One group stream of a 256-token chunk under LMC_MODEL_COUNTS, by one wave.  `tabmem` = the wave's CNT_TAB_DWORDS
// of LDS, `ring` its staging ring (ENC_RING_DWORDS), `rtab` the workgroup's copy of g_rans_rtab.
// LDSASM: the token loop requests its table entries two tokens ahead and the reciprocal one token ahead, right
// behind the step's ring store (see pass2); otherwise one token ahead, in front of it.
template <int LDSASM>
__device__ __forceinline__ void encode_group_stream_counts(const EncodeArgs& a, long long gid, u32* tabmem, u16* const ring,
                                                           const u32* rtab, int lane, PendingTile& t) {
  typedef __attribute__((address_space(3))) u32* lds_u32w;
  typedef __attribute__((address_space(3))) u16* lds_u16w;
  const int g = (int)(gid % a.G);
  const long long pc = gid / a.G;
  const int p = (int)(pc % a.P);
  const int chunk = (int)(pc / a.P);
  constexpr int Tc = (int)LMC_COUNTS_T;
  const int c = g * 64 + lane;
  const bool active = c < a.C;
  const u32* symq = a.sym4 + ((long long)chunk * a.P + p) * a.sym_stride + c;
  const bool nib = lmc_sym_nibbles((int)a.bins.b[p]);  // wave-uniform
  const u32 tab_addr = (u32)(size_t)(lds_u32w)tabmem;
  const u32 rtab_addr = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(lds_u32w) const_cast<u32*>(rtab));
  u16* const tab16 = reinterpret_cast<u16*>(tabmem);

  // ---- pass 1: histogram ----------------------------------------------------------------------------------------
  // <= 16 symbols: u32 counters [16][64] (bank = lane: conflict free); else u16 counters [32][64], lanes 2i and 2i+1
  // sharing a dword and adding 1 / 1 << 16 (as the CDF16 coder does).  Both alias the table that replaces them.
  auto pass1 = [&](auto nib_tag) {
    constexpr bool NIB = decltype(nib_tag)::value;
    constexpr int DPB = NIB ? 4 : 8;  // dwords per 32-token block
    constexpr int NB = Tc / 32;
#pragma unroll
    for (int i = 0; i < 16; i++) tabmem[i * 64 + lane] = 0;
    const u32 col = NIB ? tab_addr + 4u * (u32)lane : tab_addr + 4u * (u32)(lane >> 1);
    const u32 one = NIB ? 1u : 1u << ((lane & 1) * 16);
    u32 w[DPB], wn[DPB];
#pragma unroll
    for (int j = 0; j < DPB; j++) w[j] = active ? symq[(long long)j * a.C] : 0u;
    for (int b = 0; b < NB; b++) {
      if (b + 1 < NB) {
#pragma unroll
        for (int j = 0; j < DPB; j++) wn[j] = active ? symq[(long long)((b + 1) * DPB + j) * a.C] : 0u;
      }
      static_for<32>([&](auto itag) {
        constexpr int i = decltype(itag)::value;
        const u32 ad = row_addr_cnt<NIB, i>(w, col);
        __hip_atomic_fetch_add((lds_u32w)(size_t)ad, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      });
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = wn[j];
    }
  };
  if (nib) pass1(BoolTag<true>{});
  else pass1(BoolTag<false>{});
#if LMC_EXP_TWICE & 2  // timing experiment: the histogram pass a second time (same counts)
  wave_lds_fence();
  if (nib) pass1(BoolTag<true>{});
  else pass1(BoolTag<false>{});
#endif
  wave_lds_fence();  // every lane's ds_add has landed

  const u32 R = (u32)a.bins.b[p] - 1u;
  const BlobOff bo = lmc_blob_off((u32)a.P, (u32)Tc, (u32)a.C, (u32)a.G, (u32)a.bins.rowpre[a.P]);
  u8* const blob0 = a.blobs + (long long)chunk * a.blob_stride;
  if (g == 0) {  // one wave per (chunk, plane): checksum of the plane's scales (written by the quantise stage)
    const u32 cs = scale_checksum(reinterpret_cast<const u16*>(blob0 + bo.scales) + (long long)p * Tc, (u32)Tc, lane);
    if (lane == 0) reinterpret_cast<u32*>(blob0 + bo.scsum)[p] = cs;
  }
  // counts section of the blob (lmc_format.h): plane p = [R][C] bytes, 256 saturating to 255 -- a lane stores its own
  // channel's counts straight from its registers, one coalesced 64-byte row per symbol
  u8* const sec_row0 = blob0 + bo.cdf + (long long)a.C * a.bins.rowpre[p];  // uniform
  auto store_count = [&](u32 i, u32 v) {
    if (i < R) {  // uniform
      u8* row = sec_row0 + (long long)i * a.C;  // uniform base, lane offset c
      if (active) row[c] = (u8)min(v, 255u);
    }
  };

  // ---- table ------------------------------------------------------------------------------------------------------
  // A channel whose 256 symbols are equal is coded with count 255 and a count of 1 on symbol 0 (symbol 1 if its own
  // symbol is 0): lmc_counts_model.  Idle lanes (channel >= C) saw symbol 0 only, so they are such channels.
  u32 x = active ? LMC_COUNTS_L : 0u;
  if (nib) {
    u32 cnt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cnt[i] = tabmem[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; i++) store_count((u32)i, cnt[i]);
    u32 orv = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) orv |= cnt[i];
    if (__ballot((orv & 256u) != 0u)) {  // rare: some lane's channel is constant
      const u32 first = cnt[0] >> 8;     // 1: the constant symbol is symbol 0
#pragma unroll
      for (int i = 0; i < 16; i++) cnt[i] -= cnt[i] >> 8;
      const u32 any = orv >> 8;          // 0 or 1
      cnt[0] += any & (first ^ 1u);
      cnt[1] += any & first;
    }
    wave_lds_fence();  // the counters are dead (transposed reads above included): the table takes their place
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {  // entry = count << 23 | 2 * (symbols below): the emit threshold's upper half, start
      tabmem[i * 64 + lane] = (cnt[i] << 23) | (acc << 1);
      acc += cnt[i];
    }
  } else {
    u32 hreg[16];  // this lane's 32 counts, two per register
#pragma unroll
    for (int i = 0; i < 16; i++) hreg[i] = (u32)tab16[(2 * i) * 64 + lane] | ((u32)tab16[(2 * i + 1) * 64 + lane] << 16);
#pragma unroll
    for (int i = 0; i < 32; i++) store_count((u32)i, (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
    u32 orv = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) orv |= hreg[i];
    if (__ballot((orv & 0x01000100u) != 0u)) {
      const u32 first = (hreg[0] >> 8) & 1u;
      const u32 any = ((orv >> 8) | (orv >> 24)) & 1u;
#pragma unroll
      for (int i = 0; i < 16; i++) hreg[i] -= (hreg[i] >> 8) & 0x00010001u;
      hreg[0] += (any & (first ^ 1u)) + ((any & first) << 16);
    }
    wave_lds_fence();
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {  // entry = (symbols below) << 8 | count; 255 + 1 keeps both in a byte
      const u32 ci = (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
      tab16[i * 64 + lane] = (u16)(((acc & 0xffu) << 8) | ci);
      acc += ci;
    }
  }
  wave_lds_fence();

  // ---- pass 2: interleaved rANS, tokens 255 .. 0 -----------------------------------------------------------------
#if LMC_EXP_TWICE & 64  // timing experiment: 256 scratch slots for the whole job (stay in L2)
  u16* out = reinterpret_cast<u16*>(a.scratch + (gid % 256) * (long long)a.cap);
#else
  u16* out = reinterpret_cast<u16*>(a.scratch + gid * (long long)a.cap);
#endif
  u32 wcur = 0;  // wave-uniform word cursor
  const u32 ring_addr = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(lds_u16w)ring);
  const u64 full_exec = __builtin_amdgcn_read_exec();
  LMC_GLOBAL u32* const out32 = (LMC_GLOBAL u32*)out;
  u32 flushed = 0;  // words already in global memory (a multiple of 128), wave-uniform
  // the words of a step go to the wave's LDS ring (256 slots + a 64-slot extension: a step never wraps); whenever
  // 128 words have gathered they leave with one coalesced 256-byte store (k_encode.h: code_token)
  auto flush_ring = [&]() {
    if (wcur - flushed >= 128u) {
      wave_lds_fence();
      if (flushed & 128u) {  // the upper half leaves: bring the words that ran past slot 255 back to slots 0..
        const u32 over = wcur - flushed - 128u;  // < 64
        if ((u32)lane < over) ring[lane] = ring[ENC_RING_WORDS + lane];
      }
      (out32 + (flushed >> 1))[lane] = (reinterpret_cast<const u32*>(ring) + ((flushed & (ENC_RING_WORDS - 1)) >> 1))[lane];
      flushed += 128u;
    }
  };
  // state update x += (x / f) * (512 - f) + start, the quotient by the frequency's reciprocal {m, shc}
  auto rans_put_nib = [&](u32 e, u32 m, u32 shc) {
    u32 q;
    asm("v_mul_hi_u32 %[q], %[x], %[m]\n\t"
        "v_lshrrev_b32_sdwa %[q], %[shc], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_u32_u24 %[x], %[q], %[shc], %[x]\n\t"
        "v_add_u32_sdwa %[x], %[x], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : [x] "+v"(x), [q] "=&v"(q)
        : [m] "v"(m), [shc] "v"(shc), [e] "v"(e));
  };
  auto rans_put_byte = [&](u32 e, u32 m, u32 shc) {
    u32 q, st2;
    asm("v_mul_hi_u32 %[q], %[x], %[m]\n\t"
        "v_lshrrev_b32_sdwa %[q], %[shc], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_lshlrev_b32_sdwa %[st2], 1, %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_mad_u32_u24 %[x], %[q], %[shc], %[x]\n\t"
        "v_add_u32_e32 %[x], %[x], %[st2]"
        : [x] "+v"(x), [q] "=&v"(q), [st2] "=&v"(st2)
        : [m] "v"(m), [shc] "v"(shc), [e] "v"(e));
  };

  auto pass2 = [&](auto nib_tag) {
    constexpr bool NIB = decltype(nib_tag)::value;
    constexpr int DPB = NIB ? 4 : 8;
    constexpr int NB = Tc / 32;
    const u32 col = NIB ? tab_addr + 4u * (u32)lane : tab_addr + 2u * (u32)lane;  // LDS address of tab[0][lane]
    auto entry_at = [&](u32 ad) -> u32 {
      return NIB ? *(lds_u32w)(size_t)ad : (u32) * (lds_u16w)(size_t)ad;
    };
    auto rtab_of = [&](u32 e) -> u32x2_t {  // the reciprocal of the entry's frequency
      const u32 ra = NIB ? (e >> 20) : ((e & 0xffu) << 3);
      return *(const __attribute__((address_space(3))) u32x2_t*)(size_t)(rtab_addr + ra);
    };
    u32 w[DPB], wn[DPB];
#pragma unroll
    for (int j = 0; j < DPB; j++) w[j] = active ? LMC_SYM_LAST_LOAD(symq + (long long)((NB - 1) * DPB + j) * a.C) : 0u;
    if constexpr (LDSASM == 0) {
      // plain form: loads and waits left to the compiler, the entry and its reciprocal fetched a token ahead
      u32 e_n = entry_at(row_addr_cnt<NIB, 31>(w, col));
      u32x2_t r_n = rtab_of(e_n);
      for (int b = NB - 1; b >= 0; b--) {
#pragma unroll
        for (int j = 0; j < DPB; j++) wn[j] = (b > 0 && active) ? LMC_SYM_LAST_LOAD(symq + (long long)((b - 1) * DPB + j) * a.C) : 0u;
        static_for<32>([&](auto itag) {
          constexpr int i = 31 - decltype(itag)::value;  // token of the block, descending
          const u32 e = e_n;
          const u32x2_t r = r_n;
          if constexpr (i > 0) e_n = entry_at(row_addr_cnt<NIB, i - 1>(w, col));
          else e_n = entry_at(row_addr_cnt<NIB, 31>(wn, col));  // after the last block: row 0, read and never used
          r_n = rtab_of(e_n);
          const u32 wbase = ring_addr + ((wcur & (ENC_RING_WORDS - 1)) << 1);  // scalar
          u32 tt, cnt;
          if constexpr (NIB) {
            // emit <=> x >= count << 23: the state's upper half against the entry's upper half (count << 7)
            asm volatile("v_cmpx_ge_u32_sdwa vcc, %[x], %[e] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt)
                         : [e] "v"(e), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          } else {
            asm volatile("v_lshlrev_b32_sdwa %[t], 23, %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_cmpx_ge_u32_e32 vcc, %[x], %[t]\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt)
                         : [e] "v"(e), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          }
          wcur += cnt;
          flush_ring();
          if constexpr (NIB) rans_put_nib(e, r.x, r.y);
          else rans_put_byte(e, r.x, r.y);
        });
#pragma unroll
        for (int j = 0; j < DPB; j++) w[j] = wn[j];
      }
    } else if constexpr (LDSASM == 2) {
      // Deep form of the pipeline below: the entry of a token is requested FOUR steps ahead, its reciprocal TWO (from
      // an entry that landed two steps earlier), so the wait in front of a step's block covers requests that are two
      // steps old and leaves the previous step's three LDS operations in flight.  The one-step distance of the form
      // below is enough at 8 waves per SIMD; in the fused kernel the coding waves share their CU with workgroups that
      // fetch (k_cdf_encode at half occupancy: +28 %), and there the extra distance is what hides the LDS latency.
      u32 E0 = entry_at(row_addr_cnt<NIB, 31>(w, col));
      u32 E1 = entry_at(row_addr_cnt<NIB, 30>(w, col));
      u32 E2 = entry_at(row_addr_cnt<NIB, 29>(w, col));
      u32 E3 = entry_at(row_addr_cnt<NIB, 28>(w, col));
      u32x2_t R0 = rtab_of(E0);
      u32x2_t R1 = rtab_of(E1);
      for (int b = NB - 1; b >= 0; b--) {
#pragma unroll
        for (int j = 0; j < DPB; j++) wn[j] = (b > 0 && active) ? LMC_SYM_LAST_LOAD(symq + (long long)((b - 1) * DPB + j) * a.C) : 0u;
        static_for<32>([&](auto itag) {
          constexpr int i = 31 - decltype(itag)::value;  // token of the block, descending
          u32 ad4;  // row address of the token four further on (past the last block: row 0, read and never used)
          if constexpr (i >= 4) ad4 = row_addr_cnt<NIB, i - 4>(w, col);
          else ad4 = row_addr_cnt<NIB, 28 + i>(wn, col);
          const u32 wbase = ring_addr + ((wcur & (ENC_RING_WORDS - 1)) << 1);  // scalar
          u32 tt, cnt, ra;
          if constexpr (NIB) {
            asm volatile("v_lshrrev_b32_e32 %[ra], 20, %[e2]\n\t"
                         "v_cmpx_ge_u32_sdwa vcc, %[x], %[e0] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [ra] "=&v"(ra)
                         : [e0] "v"(E0), [e2] "v"(E2), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          } else {
            asm volatile("v_lshlrev_b32_sdwa %[ra], 3, %[e2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_lshlrev_b32_sdwa %[t], 23, %[e0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_cmpx_ge_u32_e32 vcc, %[x], %[t]\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [ra] "=&v"(ra)
                         : [e0] "v"(E0), [e2] "v"(E2), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          }
          const u32x2_t R2 = *(const __attribute__((address_space(3))) u32x2_t*)(size_t)(rtab_addr + ra);
          const u32 E4 = entry_at(ad4);
          wcur += cnt;
          flush_ring();
          if constexpr (NIB) rans_put_nib(E0, R0.x, R0.y);
          else rans_put_byte(E0, R0.x, R0.y);
          E0 = E1; E1 = E2; E2 = E3; E3 = E4;
          R0 = R1; R1 = R2;
        });
#pragma unroll
        for (int j = 0; j < DPB; j++) w[j] = wn[j];
      }
    } else {
      // Pipelined form.  On entry to a step: E0 = entry of the token to code, R0 = its reciprocal, E1 = entry of the
      // next token -- requested one (R0, E1) and two (E0) steps ago.  The step's first asm block works out the address
      // of E1's reciprocal (so the wait for E1 sits in FRONT of the block, where everything outstanding is a step
      // old) and appends the step's words; R1 = rtab[E1] and E2 = the entry of the token after next are requested
      // right behind it -- behind the ring store in the LDS queue, so the wait for them at the top of the next step
      // never waits for that store on its own account -- then the state update runs on E0 / R0.  All of the reads
      // are plain loads the compiler tracks.
      u32 E0 = entry_at(row_addr_cnt<NIB, 31>(w, col));
      u32 E1 = entry_at(row_addr_cnt<NIB, 30>(w, col));
      u32x2_t R0 = rtab_of(E0);
      for (int b = NB - 1; b >= 0; b--) {
#pragma unroll
        for (int j = 0; j < DPB; j++) wn[j] = (b > 0 && active) ? LMC_SYM_LAST_LOAD(symq + (long long)((b - 1) * DPB + j) * a.C) : 0u;
        static_for<32>([&](auto itag) {
          constexpr int i = 31 - decltype(itag)::value;  // token of the block, descending
          u32 ad2;  // row address of the token after next (past the last block: row 0, read and never used)
          if constexpr (i >= 2) ad2 = row_addr_cnt<NIB, i - 2>(w, col);
          else ad2 = row_addr_cnt<NIB, 30 + i>(wn, col);
          const u32 wbase = ring_addr + ((wcur & (ENC_RING_WORDS - 1)) << 1);  // scalar
          u32 tt, cnt, ra;
          if constexpr (NIB) {
            asm volatile("v_lshrrev_b32_e32 %[ra], 20, %[e1]\n\t"
                         "v_cmpx_ge_u32_sdwa vcc, %[x], %[e0] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [ra] "=&v"(ra)
                         : [e0] "v"(E0), [e1] "v"(E1), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          } else {
            asm volatile("v_lshlrev_b32_sdwa %[ra], 3, %[e1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_lshlrev_b32_sdwa %[t], 23, %[e0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_cmpx_ge_u32_e32 vcc, %[x], %[t]\n\t"
                         "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                         "s_nop 0\n\t"
                         "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                         "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                         "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                         "ds_write_b16 %[t], %[x]\n\t"
                         "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                         "s_mov_b64 exec, %[full]"
                         : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [ra] "=&v"(ra)
                         : [e0] "v"(E0), [e1] "v"(E1), [wb] "s"(wbase), [full] "s"(full_exec)
                         : "vcc", "scc", "memory");
          }
          const u32x2_t R1 = *(const __attribute__((address_space(3))) u32x2_t*)(size_t)(rtab_addr + ra);
          const u32 E2 = entry_at(ad2);
          wcur += cnt;
          flush_ring();
          if constexpr (NIB) rans_put_nib(E0, R0.x, R0.y);
          else rans_put_byte(E0, R0.x, R0.y);
          E0 = E1;
          E1 = E2;
          R0 = R1;
        });
#pragma unroll
        for (int j = 0; j < DPB; j++) w[j] = wn[j];
      }
    }
  };
  if (nib) pass2(BoolTag<true>{});
  else pass2(BoolTag<false>{});
#if LMC_EXP_TWICE & 4  // timing experiment: the coding pass a second time (same stream)
  wave_lds_fence();
  x = active ? LMC_COUNTS_L : 0u;
  wcur = 0;
  flushed = 0;
  asm volatile("" : "+v"(x), "+s"(wcur), "+s"(flushed) : : "memory");
  if (nib) pass2(BoolTag<true>{});
  else pass2(BoolTag<false>{});
#endif
  x = active ? x : LMC_COUNTS_L;  // idle lanes (channel >= C) carry the initial state
  // the words still in the ring (< 128 + 64)
  wave_lds_fence();
  for (u32 k = flushed + lane; k < wcur; k += 64) out[k] = ring[k & (ENC_RING_WORDS - 1)];
  // tail: states, pad, length
  out[wcur + 2 * lane] = (u16)x;
  out[wcur + 2 * lane + 1] = (u16)(x >> 16);
  wcur += 128;
  const u32 exact = wcur * 2;
  const u32 padw = ((16u - (exact & 15u)) & 15u) >> 1;
  if ((u32)lane < padw) out[wcur + lane] = 0;
  if (lane == 0 && exact + 16 > a.cap) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);
  t.chunk = chunk; t.pg = p * a.G + g; t.exact = exact; t.T = (u32)Tc; t.out = out;
}

// ---- the two-kernel path's coder launch: one wave per group stream, NW streams per workgroup ----------------
// <.., ENC_WAVES, false>: any chunk length (CDF16 or counts per stream), 4 waves and 21.5 KB of LDS per workgroup:
//     7 workgroups = 28 waves per CU.
// <.., 8, true>: launches whose chunks are all 256 tokens long (the counts model only: 4 KiB tables, one 2 KiB
//     reciprocal table shared by 8 waves): 39.9 KB per workgroup, 4 workgroups = 32 waves per CU -- the coder's time
//     falls with every wave there is to interleave.
template <bool QUADSYM, bool ENCODE, int NW = ENC_WAVES, bool COUNTS_ONLY = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 8 : 4, 8))) void k_cdf_encode(EncodeArgs a) {
  static_assert(!COUNTS_ONLY || (QUADSYM && ENCODE), "the counts coder reads the workspace and places its streams");
  constexpr int TAB_DWORDS = COUNTS_ONLY ? CNT_TAB_DWORDS : ENC_TAB_DWORDS;
  __shared__ __attribute__((aligned(16))) u32 lds_all[NW * (ENC_RING_DWORDS + TAB_DWORDS)];  // the staging rings, then the tables
  __shared__ __attribute__((aligned(16))) u32 rtab_lds[ENCODE ? RTAB_DWORDS : 4];         // counts model: reciprocals
#ifdef LMC_EXP_CDF_LDS_PAD  // timing experiment: fewer workgroups per CU (occupancy sweep of the coder)
  __shared__ u32 lds_pad[LMC_EXP_CDF_LDS_PAD];
  if (threadIdx.x == 0 && a.nchunks < 0) lds_pad[a.P] = 1;
#endif
  if (ENCODE && QUADSYM) {
    rtab_to_lds(rtab_lds);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  // everything derived from the wave id is wave-uniform: keep it in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long ngroups_total = (long long)a.nchunks * a.P * a.G;
  u32* hist = lds_all + NW * ENC_RING_DWORDS + wave * TAB_DWORDS;  // [32][64] u16 counters; lanes 2i, 2i+1 share a dword

  // Workgroup -> streams.  When the streams of a chunk fill whole workgroups, consecutive workgroups take the SAME
  // position of consecutive CHUNKS (chunk = block % nchunks): the predecessors a workgroup's placement depends on
  // (lower positions of its own chunk) were then dispatched at least nchunks workgroups earlier and have normally
  // published their lengths by the time it looks back -- with chunk-major order they finish at the same moment
  // and every look-back waits for the slowest of them.
  // ENCODE: the workgroup's work index is a ticket drawn at its start (EncodeArgs::ticket), so that the look-back
  // below only ever waits for workgroups that are running already
  u32 item = blockIdx.x;
  if constexpr (ENCODE) item = (u32)__builtin_amdgcn_readfirstlane((int)draw_ticket(a.ticket, a.ticket_base));
  long long gid = (long long)item * NW + wave;
  if (ENCODE && (a.P * a.G) % NW == 0) {
    const int wpc = a.P * a.G / NW;  // workgroups per chunk
    const int ch = (int)(item % (unsigned)a.nchunks), pos = (int)(item / (unsigned)a.nchunks);
    gid = ((long long)ch * wpc + pos) * NW + wave;
  }
  if (gid >= ngroups_total) return;
  PendingTile t;
  u16* const wring = reinterpret_cast<u16*>(lds_all + wave * (ENC_RING_DWORDS));
  if constexpr (COUNTS_ONLY) {
    encode_group_stream_counts<LMC_COUNTS_LDSASM>(a, gid, hist, wring, rtab_lds, lane, t);
  } else if constexpr (ENCODE && QUADSYM) {
    // 256-token chunks are coded on their symbol counts (LMC_MODEL_COUNTS), every other length on the 16-bit CDF
    const int chunk_of = (int)(gid / ((long long)a.P * a.G));
    const bool counts_model =
        min(a.chunk_tokens, a.tok_end - (a.tok_begin + chunk_of * a.chunk_tokens)) == (int)LMC_COUNTS_T;  // wave-uniform
    if (counts_model) encode_group_stream_counts<LMC_COUNTS_LDSASM>(a, gid, hist, wring, rtab_lds, lane, t);
    else encode_group_stream<QUADSYM, ENCODE>(a, gid, hist, wring, lane, t);
  } else {
    encode_group_stream<QUADSYM, ENCODE>(a, gid, hist, wring, lane, t);
  }
  if (!ENCODE) return;
  // ---- compaction: where does this stream go? ------------------------------------------------------------
  const int n = a.P * a.G;
  const int chunk = t.chunk;
  const u32 padded = (t.exact + 15u) & ~15u;
  unsigned long long* agg = a.agg + (long long)chunk * n;
  if (n % NW == 0) {
    // The waves of a workgroup hold consecutive streams of one chunk: they add their lengths up in LDS and
    // ONE wave runs the look-back over workgroup-level granules -- 1/NW of the granules, and of the
    // walk when a whole chunk finishes at once and nobody has an inclusive prefix yet.
    __shared__ u32 wg_len[NW];
    __shared__ u32 wg_excl;
    if (lane == 0) wg_len[wave] = padded;
    __syncthreads();
    u32 intra = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const u32 l = wg_len[w];
      intra += w < wave ? l : 0u;
      wg_total += l;
    }
    if (wave == 0) {
      const int wgi = t.pg / NW;
      if (lane == 0 && wgi > 0) agg_store(agg + wgi, AGG_A, wg_total);
      const u32 e = lookback_exclusive(agg, wgi, lane, a.status);
      if (lane == 0) {
        agg_store(agg + wgi, AGG_P, e + wg_total);
        wg_excl = e;
      }
    }
    __syncthreads();
    place_stream(a, t, wg_excl + intra, hist, lane);
  } else {
    // streams of a chunk do not fill whole workgroups: every wave publishes and looks back for itself
    if (lane == 0 && t.pg > 0) agg_store(agg + t.pg, AGG_A, padded);
    const u32 excl = lookback_exclusive(agg, t.pg, lane, a.status);
    if (lane == 0) agg_store(agg + t.pg, AGG_P, excl + padded);
    place_stream(a, t, excl, hist, lane);
  }
}
