// k_encode_counts.h -- the group-stream coder under LMC_MODEL_COUNTS (include/lmc_format.h): chunks of 2 .. 256 tokens
// (256 is the reference's chunk size, lmcache/config.py: chunk_size 256; since round 5 shorter chunks -- a ragged last
// chunk, chunk_size 128 -- are coded on the counts scaled to a sum of 256, lmc_counts_model).  Stands where
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
//
// Format v6 splits a stream's life in two, because a counts-model stream is PLACED before it is coded:
//   counts_hist_stream   pass 1: histogram -> the lane's stored counts (packed, four to a register), their OR over the
//                        lanes (the widths of the stream's head) and the stream's allocation: an upper bound of its
//                        length from the counts (lmc_counts_bits);
//   counts_open_stream   the head (k_head.h) to the stream's place, the coder's table from the counts;
//   counts_code_stream   pass 2: interleaved rANS, the words leaving for `out` -- the stream's final place in the
//                        blob (k_fused.h, the normal case) or a scratch slot.
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
// lmc_counts_bits (lmc_format.h), the stream-length bound's table: read once per symbol and stream, from the
// workgroup's LDS copy behind the reciprocals (u16 entries)
#define BITS_DWORDS 132  // 257 u16, rounded up to 16 bytes
// ... and lmc_counts_bpo behind it, the per-occurrence table of chunks below 256 tokens (round 5)
#define RTAB_LDS_DWORDS (RTAB_DWORDS + 2 * BITS_DWORDS)
struct CountsBits {
  u32 v[2 * BITS_DWORDS];
};
constexpr CountsBits make_counts_bits() {
  constexpr u16 t[257] = {LMC_COUNTS_BITS_LIST};
  constexpr u16 o[257] = {LMC_COUNTS_BPO_LIST};
  CountsBits r{};
  for (int i = 0; i < 257; i++) r.v[i >> 1] |= (u32)t[i] << (16 * (i & 1));
  for (int i = 0; i < 257; i++) r.v[BITS_DWORDS + (i >> 1)] |= (u32)o[i] << (16 * (i & 1));
  return r;
}
__device__ const CountsBits g_counts_bits = make_counts_bits();
// the counts coder's staging buffer per wave: 128 + 2 x 64 words (counts_code_stream)
#define CNT_RING_DWORDS 128

// every thread of the workgroup takes part; the caller synchronises before the first use
__device__ __forceinline__ void rtab_to_lds(u32* rtab_lds) {
  for (u32 i = threadIdx.x; i < RTAB_DWORDS; i += blockDim.x) rtab_lds[i] = g_rans_rtab.v[i];
  for (u32 i = threadIdx.x; i < 2 * BITS_DWORDS; i += blockDim.x) rtab_lds[RTAB_DWORDS + i] = g_counts_bits.v[i];
}

// the coding pass reads a symbol dword for the last time: the load says so (non-temporal), so that what is touched
// once does not push what is still needed out of L2 / Infinity Cache (same box: 1.017-1.027 -> 0.968-0.975 ms together
// with the non-temporal raw-KV loads and stream stores of the fused kernel)
#define LMC_SYM_LAST_LOAD(p) __builtin_nontemporal_load((const LMC_GLOBAL u32*)(p))

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
// ALIGNED: the wave's table slice starts at a multiple of 4 KiB (the 8-wave layouts: k_encode_fused, the counts-only
// k_cdf_encode), so bits 8 .. 11 of a lane's column address are zero and the one nibble of a dword that already
// sits there is merged in by ONE v_and_or_b32.
template <bool NIB, int I, bool ALIGNED = false>
__device__ __forceinline__ u32 row_addr_cnt(const u32* w, u32 base) {
  if constexpr (NIB) {
    constexpr int POS = 8 * (I & 3) + 4 * ((I >> 2) & 1);
    if constexpr (ALIGNED && POS == 8) {
      u32 r;
      asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w[I >> 3]), "s"(0xf00u), "v"(base));
      return r;
    } else if constexpr (ALIGNED && POS > 8) {
      // (round 5) the field moved DOWN to bits 8 .. 11 by a v_lshrrev -- the fast class, 1.05 ns against 1.85 for v_bfe --
      // then merged into the aligned column address: one fast + one normal instruction instead of two normal ones
      u32 r;
      asm("v_lshrrev_b32_e32 %0, %2, %1\n\tv_and_or_b32 %0, %0, %3, %4" : "=&v"(r) : "v"(w[I >> 3]), "n"(POS - 8), "s"(0xf00u), "v"(base));
      return r;
    } else {
      return row_addr_sh<POS, 4, 8>(w[I >> 3], base);
    }
  } else {
    constexpr int POS = 8 * (I & 3);
    if constexpr (ALIGNED && POS >= 8) {
      // rows of 128 B: symbol << 7.  Symbols are < 32, so the field is bits 7 .. 11 -- below the 4 KiB alignment of the slice
      u32 r;
      asm("v_lshrrev_b32_e32 %0, %2, %1\n\tv_and_or_b32 %0, %0, %3, %4" : "=&v"(r) : "v"(w[I >> 2]), "n"(POS - 7), "s"(0xf80u), "v"(base));
      return r;
    } else {
      return row_addr_sh<POS, 8, 7>(w[I >> 2], base);
    }
  }
}


// One group stream of a chunk of T <= 256 tokens, as its wave sees it.
struct CountsStream {
  int chunk, p, g, c;
  int T;             // tokens of the chunk (wave-uniform; a ragged last chunk has fewer than a.chunk_tokens)
  bool active, nib;  // nib is wave-uniform
  const u32* symq;   // this lane's column of the plane-chunk's symbol workspace
  u32 R;             // symbols the plane's quantiser can emit
};
// (chunk, p, g wave-uniform.  No division here: a 64-bit gid / G costs more than a hundred instructions.)
__device__ __forceinline__ CountsStream counts_stream_of(const EncodeArgs& a, int chunk, int p, int g, int lane) {
  CountsStream s;
  s.g = g;
  s.p = p;
  s.chunk = chunk;
  s.c = s.g * 64 + lane;
  s.T = min(a.chunk_tokens, a.tok_end - (a.tok_begin + chunk * a.chunk_tokens));
  s.active = s.c < a.C;
  s.symq = a.sym4 + ((long long)s.chunk * a.P + s.p) * a.sym_stride + s.c;
  s.nib = lmc_sym_nibbles((int)a.bins.b[s.p]);
  s.R = (u32)a.bins.b[s.p] - 1u;
  return s;
}

// A stream between its two passes: the lane's STORED counts (lmc_format.h: a count of 256 reads 255; a lane without
// a channel stores 0), one byte per symbol -- symbol i = byte i % 4 of pk[i / 4], <= 16-symbol planes use pk[0 .. 4) --,
// their OR over the lanes (wave-uniform: the widths of the head's fields) and the head's size.
struct CountsState {
  u32 pk[8];
  u32 wor[8];
  u32 head;
};

// VOP2 with an SDWA byte select on the second operand: D = OP(a, byte K of b) -- a count leaves its packed register
// inside the instruction that uses it.
#define LMC_SDWA_BYTE_OP(NAME, OP)                                                                                      \
  template <int K>                                                                                                      \
  __device__ __forceinline__ u32 NAME(u32 a, u32 b) {                                                                   \
    u32 d;                                                                                                              \
    if constexpr (K == 0) asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b)); \
    else if constexpr (K == 1) asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b)); \
    else if constexpr (K == 2) asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(a), "v"(b)); \
    else asm(OP " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b)); \
    return d;                                                                                                           \
  }
LMC_SDWA_BYTE_OP(sdwa_byte_shl, "v_lshlrev_b32_sdwa")  // byte K of b << a
LMC_SDWA_BYTE_OP(sdwa_byte_add, "v_add_u32_sdwa")      // a + byte K of b
LMC_SDWA_BYTE_OP(sdwa_byte_or, "v_or_b32_sdwa")        // a | byte K of b

// A lane's MODEL counts (lmc_counts_model) from its stored counts: themselves, but a channel whose 256 symbols are
// equal (stored 255, a unit missing from the sum) is coded with 255 and a count of 1 on symbol 0 (on symbol 1 if its
// own symbol is 0); a lane without a channel is coded like a constant channel of symbol 0 (it never emits).
// Packed as the stored counts are: mp[k] = four model counts (the fix-ups touch symbols 0 and 1 only, and neither
// carries out of its byte: a count that grows was 0).
template <int NS>
__device__ __forceinline__ void counts_model_pk(const u32 (&pk)[8], bool active, u32 T, u32 (&mp)[NS / 4]) {
  if (T == LMC_COUNTS_T) {  // (wave-uniform)
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < NS / 4; k++) sum = __builtin_amdgcn_sad_u8(pk[k], 0u, sum);
    const u32 deficit = LMC_COUNTS_T - sum;  // 0, or 1 for a constant channel (256 for a lane without a channel)
    const bool first = (pk[0] & 0xffu) == 255u;
#pragma unroll
    for (int k = 0; k < NS / 4; k++) mp[k] = pk[k];
    mp[0] += first ? deficit << 8 : deficit;
  } else {
    // chunks below 256 tokens (round 5): the counts scaled to a sum of 256 along the cumulative sum,
    // n[s] = floor(256 C_s / T) - floor(256 C_(s-1) / T) (lmc_format.h: lmc_counts_model; one multiply-high per symbol,
    // exact for C <= T <= 256); a channel with a single symbol comes out at 256: 255, and a count of 1 on symbol 0 / 1.
    // Per stream, not per token: plain code.
    const u32 magic = lmc_counts_scale_magic_dev(T);
    u32 cum = 0, prev = 0, big = 0, first = 0;
#pragma unroll
    for (int k = 0; k < NS / 4; k++) mp[k] = 0;
    static_for<NS>([&](auto itag) {
      constexpr int i = decltype(itag)::value;
      cum += (pk[i >> 2] >> (8 * (i & 3))) & 0xffu;
      const u32 now = __umulhi(cum << 8, magic);
      u32 n = now - prev;
      prev = now;
      if (n >= LMC_COUNTS_T) { big = 1u; first = i == 0 ? 1u : 0u; n = LMC_COUNTS_T - 1u; }
      mp[i >> 2] |= n << (8 * (i & 3));
    });
    if (big) mp[0] += first ? 1u << 8 : 1u;
  }
  if (!active) mp[0] = 0x01ffu;  // (its stored counts are all 0)
}
// The table of a <= 16-symbol plane from this lane's MODEL counts: entry = count << 23 | 2 * (symbols below) -- the
// emit threshold's upper half, and the start.  Three instructions per symbol.
__device__ __forceinline__ void counts_table_nib_pk(const u32 (&mp)[4], u32* tabmem, int lane) {
  u32 acc = 0;
  static_for<16>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    const u32 t = sdwa_byte_shl<i & 3>(23u, mp[i >> 2]);
    tabmem[i * 64 + lane] = (acc << 1) | t;
    acc = sdwa_byte_add<i & 3>(acc, mp[i >> 2]);
  });
}
// ... of a plane with more symbols: entry = (symbols below) << 8 | count in 16 bits; 255 + 1 keeps both in a byte
__device__ __forceinline__ void counts_table_byte_pk(const u32 (&mp)[8], u16* tab16, int lane) {
  u32 acc = 0;
  static_for<32>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    tab16[i * 64 + lane] = (u16)sdwa_byte_or<i & 3>(acc << 8, mp[i >> 2]);
    acc = sdwa_byte_add<i & 3>(acc, mp[i >> 2]);
  });
}

// ---- pass 1 --------------------------------------------------------------------------------------------------------
// Histogram of the stream's 256 symbols in the wave's table slice `tabmem` (<= 16 symbols: u32 counters [16][64],
// bank = lane: conflict free; else u16 counters [32][64], lanes 2i and 2i+1 sharing a dword and adding 1 / 1 << 16).
// Leaves the stream's CountsState and returns its allocation (lmc_format.h, v6): head + the bound of the words the
// active lanes can emit -- S = sum of lmc_counts_bits over a lane's model counts (`bits` = the workgroup's LDS copy),
// lmc_counts_lane_words(S) words per lane -- + the 64 states.  Wave g == 0 also writes the checksum of the plane's
// scales.  The slice is free again on return.
// PLANE (round 6, k_fused.h): the counters are already there -- the plane's histogram was taken while the plane was
// quantised, into the workgroup's eight table slices at LDS address 0 (k_fused.h: quantize_oct_hist has the layout):
// the lane reads its channel's 16 / 32 counters instead of 32 / 64 workspace dwords, and its slice is not touched.
template <bool ALIGNED = false, bool PLANE = false>
__device__ __forceinline__ u32 counts_hist_stream(const EncodeArgs& a, const CountsStream& s, u32* tabmem,
                                                  const u32* bits,
                                                  int lane, CountsState& cs, u32 vq_base = 0) {
  typedef __attribute__((address_space(3))) u32* lds_u32w;
  typedef const __attribute__((address_space(3))) u16* lds_u16p;
  typedef const __attribute__((address_space(3))) u8* lds_u8p;
  const int Tc = __builtin_amdgcn_readfirstlane(s.T);  // 2 .. 256, wave-uniform (said so: the loops below run on scalar counters)
  const u32 tab_addr = (u32)(size_t)(lds_u32w)tabmem;
  if (ALIGNED && (tab_addr & 0xfffu)) __builtin_trap();  // (wave-uniform: the layout is static)
  u16* const tab16 = reinterpret_cast<u16*>(tabmem);
  auto pass1 = [&](auto nib_tag) {
    constexpr bool NIB = decltype(nib_tag)::value;
    constexpr int DPB = NIB ? 4 : 8;  // dwords per 32-token block
    const int NB = Tc >> 5, rem = Tc & 31;                    // whole 32-token blocks, tokens of the last partial one
    const int rows = NIB ? (Tc + 7) >> 3 : (Tc + 3) >> 2;     // workspace dwords of the lane's column that hold tokens
#pragma unroll
    for (int i = 0; i < 16; i++) tabmem[i * 64 + lane] = 0;
    const u32 col = NIB ? tab_addr + 4u * (u32)lane : tab_addr + 4u * (u32)(lane >> 1);
    const u32 one = NIB ? 1u : 1u << ((lane & 1) * 16);
    u32 w[DPB], wn[DPB];
#pragma unroll
    for (int j = 0; j < DPB; j++) w[j] = (s.active && j < rows) ? s.symq[(long long)j * a.C] : 0u;
    for (int b = 0; b < NB; b++) {
      if (b + 1 < NB || rem) {
#pragma unroll
        for (int j = 0; j < DPB; j++)
          wn[j] = (s.active && (b + 1) * DPB + j < rows) ? s.symq[(long long)((b + 1) * DPB + j) * a.C] : 0u;
      }
      static_for<32>([&](auto itag) {
        constexpr int i = decltype(itag)::value;
        const u32 ad = row_addr_cnt<NIB, i, ALIGNED>(w, col);
        __hip_atomic_fetch_add((lds_u32w)(size_t)ad, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      });
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = wn[j];
    }
    // the partial block of a chunk whose length is no multiple of 32 (a ragged last chunk, chunk_size 236 ...): a plain
    // loop, the token's dword picked by wave-uniform selects
    for (int i = 0; i < rem; i++) {
      const int j = NIB ? i >> 3 : i >> 2;
      u32 wd = w[0];
#pragma unroll
      for (int k = 1; k < DPB; k++) wd = j == k ? w[k] : wd;
      const u32 sym = NIB ? (wd >> (8 * (i & 3) + 4 * ((i >> 2) & 1))) & 15u : (wd >> (8 * (i & 3))) & 0xffu;
      const u32 ad = col + (sym << (NIB ? 8 : 7));
      __hip_atomic_fetch_add((lds_u32w)(size_t)ad, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  };
  // PLANE: the channel of this lane, c = 64 g + lane, was quantised by lane qx = (c % 512) / 8 as element e = c % 8 of
  // its channel run it = c / 512
  // (k_hist.h: virtual quantising lane vq = vq_base + c / 8 with c = 64 g + lane the channel within its plane; vq_base = 0
  // for a plane of up to 1024 channels, pj * GL for plane pj of an item of narrow planes)
  const u32 vq = vq_base + (((u32)s.g) << 3) + ((u32)lane >> 3);
  const u32 qx = vq & 63u, qe = (u32)lane & 7u, qit = vq >> 6;
  u32 ph = 0;  // LDS byte address of this lane's counter of symbol 0 (rows of 256 B)
  if constexpr (PLANE) {
    if (s.nib) {
      ph = (qit * 4u + (qe >> 1)) * 4096u + 4u * qx + 2u * (qe & 1u);
    } else {
      ph = (qit * 2u + (qe >> 2)) * 8192u + 4u * qx + (qe & 3u);
      // token 0 of the chunk was left out of the counters (so that none of them can reach 256 and carry into its
      // neighbour): add it now, unless its counter stands at 255 -- a constant channel, whose count is stored as 255 anyway
      const u32 sym0 = s.active ? (s.symq[0] & 0xffu) : 0u;
      const u32 a0 = ph + (sym0 << 8);
      const u32 c0 = (u32) * (lds_u8p)(size_t)a0;
      if (s.active && c0 != 255u)
        __hip_atomic_fetch_add((lds_u32w)(size_t)(a0 & ~3u), 1u << (8u * (a0 & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
    if (s.nib) pass1(BoolTag<true>{});
    else pass1(BoolTag<false>{});
  }
  wave_lds_fence();  // every lane's ds_add has landed

  if (s.g == 0) {  // one wave per (chunk, plane): checksum of the plane's scales (written by the quantise stage)
    const BlobOff bo = lmc_blob_off((u32)a.P, (u32)Tc, (u32)a.G);
    u8* const blob0 = a.blobs + (long long)s.chunk * a.blob_stride;
    const u32 sum = scale_checksum(reinterpret_cast<const u16*>(blob0 + bo.scales) + (long long)s.p * Tc, (u32)Tc, lane);
    if (lane == 0) reinterpret_cast<u32*>(blob0 + bo.scsum)[s.p] = sum;
  }
  // the lane's stored counts: min(count, 255), four to a register; a lane without a channel (it saw symbol 0 only)
  // stores nothing
  // (min and byte insert are ONE instruction: v_min_u32_sdwa writes byte i % 4 of the register and keeps the others)
#pragma unroll
  for (int k = 0; k < 8; k++) cs.pk[k] = 0u;
  const u32 cap = s.active ? 255u : 0u;
  auto put = [&](auto itag, u32 c) {
    constexpr int i = decltype(itag)::value;
    if constexpr ((i & 3) == 0)
      asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(cs.pk[i >> 2]) : "v"(c), "v"(cap));
    else if constexpr ((i & 3) == 1)
      asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(cs.pk[i >> 2]) : "v"(c), "v"(cap));
    else if constexpr ((i & 3) == 2)
      asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(cs.pk[i >> 2]) : "v"(c), "v"(cap));
    else
      asm("v_min_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(cs.pk[i >> 2]) : "v"(c), "v"(cap));
  };
  if constexpr (PLANE) {
    if (s.nib) static_for<16>([&](auto itag) { put(itag, (u32) * (lds_u16p)(size_t)(ph + 256u * decltype(itag)::value)); });
    else static_for<32>([&](auto itag) { put(itag, (u32) * (lds_u8p)(size_t)(ph + 256u * decltype(itag)::value)); });
  } else {
    if (s.nib) static_for<16>([&](auto itag) { put(itag, tabmem[decltype(itag)::value * 64 + lane]); });
    else static_for<32>([&](auto itag) { put(itag, (u32)tab16[decltype(itag)::value * 64 + lane]); });
  }
  wave_lds_fence();  // the counters are dead: every lane holds its counts
  head_or_counts<8>(cs.pk, cs.wor);
  cs.head = head_bytes_of<8, 8>(cs.wor, s.R);
  // the bound: S over the lane's MODEL counts = its stored counts, plus the count of 1 a constant channel's model
  // gives a second symbol (stored 255 alone: the sum is a unit short)
  // (`bits` points into a __shared__ array of the kernel: a constant after inlining, so it rides in the
  // loads' offset field and a symbol costs a shift out of its byte -- one SDWA instruction -- and an add)
  __builtin_assume(bits != nullptr);  // (else the cast below carries a null check and is no constant)
  const u32 bits_addr = (u32)(size_t)(lds_u16p) reinterpret_cast<const u16*>(bits);
  u32 S = 0;
  auto bound = [&](auto ns_tag) {
    constexpr int NS = decltype(ns_tag)::value;
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < NS / 4; k++) sum = __builtin_amdgcn_sad_u8(cs.pk[k], 0u, sum);
    // (read unconditionally: a load under a condition becomes a branch, and every 16-bit value that crosses it costs
    // a v_and)
    static_for<NS>([&](auto itag) {
      constexpr int i = decltype(itag)::value;
      u32 c2;  // 2 * count of symbol i
      if constexpr ((i & 3) == 0) asm("v_lshlrev_b32_sdwa %0, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(c2) : "v"(cs.pk[i >> 2]));
      else if constexpr ((i & 3) == 1) asm("v_lshlrev_b32_sdwa %0, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(c2) : "v"(cs.pk[i >> 2]));
      else if constexpr ((i & 3) == 2) asm("v_lshlrev_b32_sdwa %0, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(c2) : "v"(cs.pk[i >> 2]));
      else asm("v_lshlrev_b32_sdwa %0, 1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(c2) : "v"(cs.pk[i >> 2]));
      S += (u32) * (lds_u16p)(size_t)(c2 + bits_addr);
    });
    const u32 bits_of_1 = (u32) * (lds_u16p)(size_t)(bits_addr + 2u);
    S += sum == LMC_COUNTS_T - 1u ? bits_of_1 : 0u;
  };
  // chunks below 256 tokens: S = sum of count * lmc_counts_bpo[model count] (lmc_format.h), the model counts worked out
  // as counts_model_pk does
  auto bound_T = [&](auto ns_tag) {
    constexpr int NS = decltype(ns_tag)::value;
    const u32 magic = lmc_counts_scale_magic_dev((u32)Tc);
    const u32 bpo_addr = bits_addr + 4u * BITS_DWORDS;
    u32 cum = 0, prev = 0;
    static_for<NS>([&](auto itag) {
      constexpr int i = decltype(itag)::value;
      const u32 c = (cs.pk[i >> 2] >> (8 * (i & 3))) & 0xffu;
      cum += c;
      const u32 now = __umulhi(cum << 8, magic);
      const u32 n = min(now - prev, LMC_COUNTS_T - 1u);
      prev = now;
      S += c * (u32) * (lds_u16p)(size_t)(bpo_addr + 2u * n);
    });
  };
  if (Tc == (int)LMC_COUNTS_T) {
    if (s.nib) bound(IntTag<16>{});
    else bound(IntTag<32>{});
  } else {
    if (s.nib) bound_T(IntTag<16>{});
    else bound_T(IntTag<32>{});
  }
  const u32 lw = s.active ? (S + 6u * ((S >> 12) + 2u)) >> 12 : 0u;  // lmc_counts_lane_words
  const u32 words = (u32)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(lw));
  return (cs.head + 2u * (words + 128u) + 15u) & ~15u;  // lmc_counts_alloc_bytes
}

// The stream opens: its head to `out` (16-byte aligned: the stream's place in the blob, or a scratch slot), the coder's
// table to the wave's slice (whose previous contents are dead).  The words go to out + cs.head.
__device__ __forceinline__ void counts_open_stream(const CountsStream& s, const CountsState& cs, u8* out, u32* tabmem, int lane) {
  (void)head_write<8, 8>(out, cs.pk, cs.wor, s.R, tabmem, lane);  // (the slice is idle: the table comes next)
  wave_lds_fence();
  if (s.nib) {
    u32 mp[4];
    counts_model_pk<16>(cs.pk, s.active, (u32)s.T, mp);
    counts_table_nib_pk(mp, tabmem, lane);
  } else {
    u32 mp[8];
    counts_model_pk<32>(cs.pk, s.active, (u32)s.T, mp);
    counts_table_byte_pk(mp, reinterpret_cast<u16*>(tabmem), lane);
  }
  wave_lds_fence();
}

// ---- pass 2: interleaved rANS, tokens T - 1 .. 0 -----------------------------------------------------------------
// `tabmem` holds the stream's table, `ring` is the wave's staging buffer (CNT_RING_DWORDS = 256 words, used linearly:
// see below), `rtab` the workgroup's copy of g_rans_rtab.  The words of a step go to the buffer behind the words still
// waiting in it; whenever its first 128 words are complete they leave with one coalesced 256-byte store to `out`
// (wave-uniform) and the rest moves down.  NT: the stores are non-temporal (the stream is at its final place: nobody
// reads it again) -- the callers that code in place are the 8-wave layouts, whose table slices are 4 KiB aligned
// (row_addr_cnt's ALIGNED).  The chunk's whole 32-token blocks run through the unrolled, software-pipelined loop; the
// <= 31 tokens behind them (chunk lengths that are no multiple of 32) through a plain loop in front of it.
// Returns the exact length in bytes of words + states (the head in front of `out` not counted); the 64 states follow
// the words, then zeros up to a multiple of 16 (`out` is 16-byte aligned).
template <bool NT>
__device__ __forceinline__ u32 counts_code_stream(const EncodeArgs& a, const CountsStream& s, u32* tabmem, u16* const ring,
                                                  const u32* rtab, int lane, u16* const out_v) {
  typedef __attribute__((address_space(3))) u32* lds_u32w;
  typedef __attribute__((address_space(3))) u16* lds_u16w;
  const int Tc = __builtin_amdgcn_readfirstlane(s.T);  // 2 .. 256, wave-uniform (said so: the loops below run on scalar counters)
  const u32 tab_addr = (u32)(size_t)(lds_u32w)tabmem;
  if (NT && (tab_addr & 0xfffu)) __builtin_trap();  // (wave-uniform: the layout is static)
  const u32 rtab_addr = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(lds_u32w) const_cast<u32*>(rtab));
  u32 x = s.active ? LMC_COUNTS_L : 0u;  // idle lanes stay at 0 and never emit
  const u32 ring_addr = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(lds_u16w)ring);
  const u64 full_exec = __builtin_amdgcn_read_exec();
  u16* const out = reinterpret_cast<u16*>(uniform_ptr64(out_v));
  // The staging buffer is LINEAR (round 5; it was a ring whose cursor was masked, shifted and re-based every token):
  // `wb` is the LDS byte address of the next free slot, a step's words go to wb + 2 * rank and wb += 2 * count -- one
  // scalar instruction -- and every SECOND token one compare asks whether the first 128 words are complete.  If so
  // they leave with one coalesced 256-byte store, the < 128 words behind them move down to the buffer's start and
  // wb steps back by 256 bytes.  Two steps add at most 128 words to fewer than 128: 256 words = 512 B of the wave's
  // CNT_RING_DWORDS.  What this removes from every token step: s_sub / s_cmpk / a TAKEN s_cbranch (now every second
  // step), s_lshl / s_and / s_add of the cursor -- 3.3 + 0.9 ns of the step's 24.2 in the issue-slot replica
  // (tools/probes/issue_model.py, profiles/r05_issue_model.md).
  // The 256-byte pieces leave ALIGNED to 256 bytes (round 6): a stream's words begin at a multiple of 16 bytes, and
  // pieces cut from there straddle a 64-byte write granule at either end three times out of four -- the streams are
  // stored non-temporal, the two halves of such a granule reach the fabric as two partial writes (WRITE_SIZE of the
  // fused kernel: 0.557 GB for 0.464 GB of blob = the 4.75 / 4 granules per piece this predicts).  So the buffer starts
  // out `pre` bytes full (pre = the distance of the first word from the 256-byte boundary below it: slots nobody
  // writes), the descriptor's base is that boundary, and the FIRST piece leaves without its first `pre` bytes; every
  // later piece is two whole 128-byte lines.  Same bytes in the blob.
#ifndef LMC_FLUSH_ALIGN
#define LMC_FLUSH_ALIGN 1
#endif
  const u32 pre = LMC_FLUSH_ALIGN ? (u32)__builtin_amdgcn_readfirstlane((int)((u32)(size_t)out & 255u)) : 0u;  // a multiple of 16 (the caller's contract)
  u32 skip = pre;      // bytes at the front of the next piece that are not the stream's (pre until the first piece has left)
  u32 wb = ring_addr + pre;
  const u32 wlimit = ring_addr + 256u;
  u32 flushed = 0;     // words already in global memory (a multiple of 128, counted from the boundary), wave-uniform
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((u8*)out - pre), (short)0, (int)0xfffffff0u, 0x00020000);
  auto flush_ring = [&]() {
#ifndef LMC_FLUSH_LIKELY
#define LMC_FLUSH_LIKELY 0  // 0: the flush is laid out behind the loop (the common path falls through an untaken branch)
#endif
    if (__builtin_expect(wb >= wlimit, LMC_FLUSH_LIKELY)) {
      wave_lds_fence();
      // (the lane's slot address is worked out HERE, behind an opaque asm: hoisted out of the token loop it becomes one
      // more live register, which the 64-VGPR kernels spill -- and its reload in this block waits for vmcnt(0), i.e.
      // for the symbol prefetch and the previous flush's store: measured +4 % on k_cdf_encode)
      u32 lane_here = (u32)lane;
      asm volatile("" : "+v"(lane_here));  // (opaque: what follows cannot be computed outside this block)
      const u32 slot_addr = ring_addr + 4u * lane_here;
      const u32 v = *(lds_u32w)(size_t)slot_addr;
      const u32 up = *(lds_u32w)(size_t)(slot_addr + 256u);  // words 128 .. 255: whatever of them is in use moves down
      // (uniform base in the descriptor, the lane in the vector offset, the stream position in the scalar offset)
      const u32 voff = 4u * lane_here;
#ifndef LMC_EXP_ENC_NO_FLUSH  // (timing experiment: the coder without its stream stores; blobs are wrong)
      if (!LMC_FLUSH_ALIGN || voff >= skip) __builtin_amdgcn_raw_buffer_store_b32((int)v, out_rsrc, (int)voff, (int)(flushed << 1), NT ? 2 : 0);
      skip = 0;
#else
      asm volatile("" :: "v"(v), "v"(voff));
#endif
      if (voff < wb - wlimit) *(lds_u32w)(size_t)slot_addr = up;  // (a half-used last dword brings a stale upper half along: the next word overwrites it)
      flushed += 128u;
      wb -= 256u;
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
  auto rans_put_byte = [&](_Float16 e, u32 m, u32 shc) {
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
    const int NB = Tc >> 5, rem = Tc & 31;                 // whole 32-token blocks; tokens of the partial block behind them
    const int rows = NIB ? (Tc + 7) >> 3 : (Tc + 3) >> 2;  // workspace dwords of the lane's column that hold tokens
    const u32 col = NIB ? tab_addr + 4u * (u32)lane : tab_addr + 2u * (u32)lane;  // LDS address of tab[0][lane]
    // A byte plane's 16-bit entries travel as _Float16: the blocks below only ever take them apart by SDWA byte
    // selects, so the register's upper half does not matter -- but an integer of 16 bits that lives across the
    // step's branches is zero-extended where it is used (one v_and per step); a half is passed as it is.
    using ET = std::conditional_t<NIB, u32, _Float16>;
    auto entry_at = [&](u32 ad) -> ET {
      if constexpr (NIB) return *(lds_u32w)(size_t)ad;
      else return __builtin_bit_cast(_Float16, *(lds_u16w)(size_t)ad);
    };
    auto rtab_of = [&](ET e) -> u32x2_t {  // the reciprocal of the entry's frequency
      u32 ra;
      if constexpr (NIB) ra = e >> 20;
      else ra = ((u32)__builtin_bit_cast(u16, e) & 0xffu) << 3;
      return *(const __attribute__((address_space(3))) u32x2_t*)(size_t)(rtab_addr + ra);
    };
    // the word append of one step: under exec = emitting lanes -- v_cmpx on the state's upper half against the entry's
    // (count << 7), mbcnt rank, ds_write_b16 into the staging buffer, x >>= 16, exec restored; also works out the
    // address of the reciprocal of entry e2 (the main loop's pipeline) and moves the buffer's write address on:
    // wb += 2 * words.  v_mbcnt reads the VCC that v_cmpx wrote: gfx950 wants two wait states in between (the compiler
    // puts `s_nop 1` there), and the two scalar instructions the step needs anyway -- the word count and the new write
    // address (into a second register: the append still uses the old one) -- are those two.  (Until round 5 an s_nop 0
    // stood where the s_lshl1_add_u32 stands now and the address was moved on behind the block: one scalar instruction
    // more per token.)
    auto emit = [&](ET e0, ET e2, u32& ra) {
      u32 tt, cnt, wbn;
      if constexpr (NIB) {
        asm volatile("v_lshrrev_b32_e32 %[ra], 20, %[e2]\n\t"
                     "v_cmpx_ge_u32_sdwa vcc, %[x], %[e0] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                     "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                     "s_lshl1_add_u32 %[wbn], %[cnt], %[wb]\n\t"
                     "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                     "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                     "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                     "ds_write_b16 %[t], %[x]\n\t"
                     "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                     "s_mov_b64 exec, %[full]"
                     : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [wbn] "=&s"(wbn), [ra] "=&v"(ra)
                     : [e0] "v"(e0), [e2] "v"(e2), [wb] "s"(wb), [full] "s"(full_exec)
                     : "vcc", "scc", "memory");
      } else {
        asm volatile("v_lshlrev_b32_sdwa %[ra], 3, %[e2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                     "v_lshlrev_b32_sdwa %[t], 23, %[e0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                     "v_cmpx_ge_u32_e32 vcc, %[x], %[t]\n\t"
                     "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                     "s_lshl1_add_u32 %[wbn], %[cnt], %[wb]\n\t"
                     "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                     "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                     "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                     "ds_write_b16 %[t], %[x]\n\t"
                     "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                     "s_mov_b64 exec, %[full]"
                     : [x] "+v"(x), [t] "=&v"(tt), [cnt] "=&s"(cnt), [wbn] "=&s"(wbn), [ra] "=&v"(ra)
                     : [e0] "v"(e0), [e2] "v"(e2), [wb] "s"(wb), [full] "s"(full_exec)
                     : "vcc", "scc", "memory");
      }
      wb = wbn;
    };
    u32 w[DPB], wn[DPB];
    // the last whole block (the first one coded), and in wn the partial block behind it
#pragma unroll
    for (int j = 0; j < DPB; j++) {
      w[j] = (NB > 0 && s.active) ? LMC_SYM_LAST_LOAD(s.symq + (long long)((NB - 1) * DPB + j) * a.C) : 0u;
      wn[j] = (rem && s.active && NB * DPB + j < rows) ? LMC_SYM_LAST_LOAD(s.symq + (long long)(NB * DPB + j) * a.C) : 0u;
    }
    // ---- the partial block of a chunk whose length is no multiple of 32 (tokens Tc - 1 .. 32 NB; round 5): a plain
    // loop -- the token's dword by wave-uniform selects, entry and reciprocal fetched where they are needed, the buffer
    // tested after every token.  At most 31 of a chunk's tokens come through here.
    for (int i = rem - 1; i >= 0; i--) {
      const int j = NIB ? i >> 3 : i >> 2;
      u32 wd = wn[0];
#pragma unroll
      for (int k = 1; k < DPB; k++) wd = j == k ? wn[k] : wd;
      const u32 sym = NIB ? (wd >> (8 * (i & 3) + 4 * ((i >> 2) & 1))) & 15u : (wd >> (8 * (i & 3))) & 0xffu;
      const ET E = entry_at(col + (sym << (NIB ? 8 : 7)));
      const u32x2_t R = rtab_of(E);
      u32 ra_unused;
      emit(E, E, ra_unused);
      flush_ring();
      if constexpr (NIB) rans_put_nib(E, R.x, R.y);
      else rans_put_byte(E, R.x, R.y);
    }
#ifdef LMC_EXP_SKIP_TOKEN_LOOP  // instruction-count experiments (tools/scripts/quick_pmc.sh): everything but pass 2's token loop
    return;
#endif
    if (NB == 0) return;
    // The table pipeline: the entry of a token is requested FOUR steps ahead, its reciprocal TWO (from an entry that
    // landed two steps earlier), so the wait in front of a step's block covers requests that are two steps old and
    // leaves the previous step's three LDS operations in flight.  The step's asm block works out the address of the
    // reciprocal and appends the step's words -- under exec = emitting lanes: v_cmpx on the state's upper half against
    // the entry's (count << 7), mbcnt rank, ds_write_b16 into the staging buffer, x >>= 16, exec restored (`emit` above) --
    // and the loads are plain loads the compiler tracks, issued right behind the block (behind the buffer store in the
    // LDS queue).
    ET E0 = entry_at(row_addr_cnt<NIB, 31, NT>(w, col));
    ET E1 = entry_at(row_addr_cnt<NIB, 30, NT>(w, col));
    ET E2 = entry_at(row_addr_cnt<NIB, 29, NT>(w, col));
    ET E3 = entry_at(row_addr_cnt<NIB, 28, NT>(w, col));
    u32x2_t R0 = rtab_of(E0);
    u32x2_t R1 = rtab_of(E1);
    for (int b = NB - 1; b >= 0; b--) {
#pragma unroll
      for (int j = 0; j < DPB; j++) wn[j] = (b > 0 && s.active) ? LMC_SYM_LAST_LOAD(s.symq + (long long)((b - 1) * DPB + j) * a.C) : 0u;
      static_for<32>([&](auto itag) {
        constexpr int i = 31 - decltype(itag)::value;  // token of the block, descending
        u32 ad4;  // row address of the token four further on (past the last block: row 0, read and never used)
        if constexpr (i >= 4) ad4 = row_addr_cnt<NIB, i - 4, NT>(w, col);
        else ad4 = row_addr_cnt<NIB, 28 + i, NT>(wn, col);
        u32 ra;
        emit(E0, E2, ra);  // (wb += 2 * words)
        const u32x2_t R2 = *(const __attribute__((address_space(3))) u32x2_t*)(size_t)(rtab_addr + ra);
        const ET E4 = entry_at(ad4);
#ifndef LMC_FLUSH_EVERY
#define LMC_FLUSH_EVERY 2  // tokens between two tests of the staging buffer (1 or 2: the buffer holds 128 + 2 x 64 words)
#endif
        if constexpr (LMC_FLUSH_EVERY == 1 || (i & 1) == 0) flush_ring();
        if constexpr (NIB) rans_put_nib(E0, R0.x, R0.y);
        else rans_put_byte(E0, R0.x, R0.y);
        E0 = E1; E1 = E2; E2 = E3; E3 = E4;
        R0 = R1; R1 = R2;
      });
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = wn[j];
    }
  };
  if (s.nib) pass2(BoolTag<true>{});
  else pass2(BoolTag<false>{});
  x = s.active ? x : LMC_COUNTS_L;  // idle lanes (channel >= C) carry the initial state
  // The end of the stream -- the words still in the buffer, the 64 states, zeros up to a multiple of 16 bytes -- leaves
  // through the buffer as well (round 6): the states are appended to it, and what it holds goes out as one dword per lane
  // in two pieces that continue the 256-byte grid of the pieces before.  (Until then: two 16-bit stores per lane for the
  // states, one or two for the words, one for the zeros -- a dozen partial write granules per stream, 35 MB of
  // WRITE_SIZE per 16 k context.)
  flush_ring();  // (fewer than 128 words are left, whatever test the loops ended on)
  wave_lds_fence();
  const u32 pend_b = wb - ring_addr;  // bytes in the buffer, counted from the boundary: < 256
  {
    const u32 sa = wb + 4u * (u32)lane;  // (2 mod 4 when the stream's word count is odd: two 16-bit writes)
    *(lds_u16w)(size_t)sa = (u16)x;
    *(lds_u16w)(size_t)(sa + 2u) = (u16)(x >> 16);
  }
  wave_lds_fence();
  const u32 tot_b = pend_b + 256u;            // < 512 = the buffer
  const u32 end_b = (tot_b + 15u) & ~15u;     // (pre is a multiple of 16: so is the stream's padded length)
  const u32 exact = (flushed << 1) + tot_b - pre;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const u32 voff = 256u * (u32)h + 4u * (u32)lane;
    if (voff < end_b && voff >= skip) {
      u32 v = *(lds_u32w)(size_t)(ring_addr + voff);
      v = voff >= tot_b ? 0u : (voff + 2u == tot_b ? (v & 0xffffu) : v);
      __builtin_amdgcn_raw_buffer_store_b32((int)v, out_rsrc, (int)voff, (int)(flushed << 1), NT ? 2 : 0);
    }
  }
  return exact;
}

// ---- the two-kernel path's coder launch: one wave per group stream, NW consecutive streams per workgroup ----------
// <.., ENC_WAVES, false>: any chunk length (CDF16 or counts per stream), 4 waves and 21.5 KB of LDS per workgroup:
//     7 workgroups = 28 waves per CU.
// <.., 8, true>: launches whose chunks are all 256 tokens long (the counts model only: 4 KiB tables, one 2 KiB
//     reciprocal table shared by 8 waves): 39.9 KB per workgroup, 4 workgroups = 32 waves per CU -- the coder's time
//     falls with every wave there is to interleave.
// Every stream is coded into its scratch slot and placed afterwards: the allocations (lmc_format.h, v6 -- the bound for
// a counts stream, the exact length for a CDF16 one) are prefix-summed over the chunk by a single-pass look-back, one
// granule per workgroup.  (The fused kernel, k_fused.h, runs the look-back between the two passes and codes straight
// into the blob.)
template <bool QUADSYM, bool ENCODE, int NW = ENC_WAVES, bool COUNTS_ONLY = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 8 : 4, 8))) void k_cdf_encode(EncodeArgs a) {
  static_assert(!COUNTS_ONLY || (QUADSYM && ENCODE), "the counts coder reads the workspace and places its streams");
  constexpr int TAB_DWORDS = COUNTS_ONLY ? CNT_TAB_DWORDS : ENC_TAB_DWORDS;
  constexpr int RING_DWORDS = COUNTS_ONLY ? CNT_RING_DWORDS : ENC_RING_DWORDS;  // (the CDF16 coder's ring is 256 + 64 words)
  __shared__ __attribute__((aligned(4096))) u32 lds_all[NW * (TAB_DWORDS + RING_DWORDS)];  // the tables, then the staging rings (counts-only launch: 4 KiB slices at multiples of 4 KiB, row_addr_cnt ALIGNED)
  __shared__ __attribute__((aligned(16))) u32 rtab_lds[ENCODE ? RTAB_LDS_DWORDS : 4];     // counts model: reciprocals, bound table
  if (ENCODE && QUADSYM) {
    rtab_to_lds(rtab_lds);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  // everything derived from the wave id is wave-uniform: keep it in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long ngroups_total = (long long)a.nchunks * a.P * a.G;
  u32* hist = lds_all + wave * TAB_DWORDS;  // [32][64] u16 counters; lanes 2i, 2i+1 share a dword

  // Workgroup -> streams.  When the streams of a chunk fill whole workgroups, consecutive workgroups take the SAME
  // position of consecutive CHUNKS (chunk = block % nchunks): the predecessors a workgroup's placement depends on
  // (lower positions of its own chunk) were then dispatched at least nchunks workgroups earlier and have normally
  // published their lengths by the time it looks back -- with chunk-major order they finish at the same moment
  // and every look-back waits for the slowest of them.
  // ENCODE: the workgroup's work index is a ticket drawn at its start (EncodeArgs::ticket), so that the look-back
  // below only ever waits for workgroups that are running already
  u32 item = blockIdx.x;
  if constexpr (ENCODE) item = (u32)__builtin_amdgcn_readfirstlane((int)draw_ticket(a.ticket, a.ticket_base));
  // stream = (chunk_i, pg_i), pg_i = p * G + g; all 32-bit (nchunks <= 65535, P * G <= 2^14): a 64-bit division costs
  // more than a hundred instructions
  const u32 npg = (u32)(a.P * a.G);
  u32 chunk_i, pg_i;
  if (ENCODE && npg % NW == 0) {
    chunk_i = item % (unsigned)a.nchunks;
    pg_i = (item / (unsigned)a.nchunks) * NW + (u32)wave;
  } else {
    const u32 lin = item * NW + (u32)wave;
    chunk_i = lin / npg;
    pg_i = lin - chunk_i * npg;
  }
  const long long gid = (long long)chunk_i * npg + pg_i;
  if (gid >= ngroups_total) return;
  PendingTile t;
  u16* const wring = reinterpret_cast<u16*>(lds_all + NW * TAB_DWORDS + wave * RING_DWORDS);
  u32 alloc = 0;  // the stream's allocation in the blob
  auto counts_stream = [&]() {
    const u32 p_i = pg_i / (u32)a.G;
    const CountsStream s = counts_stream_of(a, (int)chunk_i, (int)p_i, (int)(pg_i - p_i * (u32)a.G), lane);
    CountsState cs;
    alloc = counts_hist_stream(a, s, hist, rtab_lds + RTAB_DWORDS, lane, cs);
    u8* const slot = a.scratch + gid * (long long)a.cap;
    counts_open_stream(s, cs, slot, hist, lane);
    t.exact = cs.head + counts_code_stream<false>(a, s, hist, wring, rtab_lds, lane, reinterpret_cast<u16*>(slot + cs.head));
    t.chunk = s.chunk; t.pg = s.p * a.G + s.g; t.T = (u32)s.T; t.out = reinterpret_cast<const u16*>(slot);
    if (lane == 0 && (t.exact > alloc || t.exact + 16 > a.cap)) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);
  };
  if constexpr (COUNTS_ONLY) {
    // Launches of 256-token chunks only (lmc_api.hip: P * G is a multiple of NW): like the fused kernel, the streams are
    // placed BETWEEN their two passes -- histogram, allocation, one look-back per workgroup, then the coder writes
    // straight into the blob.  No scratch slot, no copy.
    const int n = a.P * a.G;
    const u32 p_i = pg_i / (u32)a.G;
    const CountsStream s = counts_stream_of(a, (int)chunk_i, (int)p_i, (int)(pg_i - p_i * (u32)a.G), lane);
    CountsState cs;
    alloc = counts_hist_stream<true>(a, s, hist, rtab_lds + RTAB_DWORDS, lane, cs);
    __shared__ u32 wg_alloc[NW];
    __shared__ u32 wg_excl;
    if (lane == 0) wg_alloc[wave] = alloc;
    __syncthreads();
    u32 before = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      before += w < wave ? wg_alloc[w] : 0u;
      wg_total += wg_alloc[w];
    }
    if (wave == 0) {
      unsigned long long* agg = a.agg + (long long)chunk_i * n;
      const int wgi = (int)pg_i / NW;
      if (lane == 0 && wgi > 0) agg_store(agg + wgi, AGG_A, wg_total);
      const u32 e = lookback_exclusive(agg, wgi, lane, a.status);
      if (lane == 0) {
        agg_store(agg + wgi, AGG_P, e + wg_total);
        wg_excl = e;
      }
    }
    __syncthreads();
    const u32 beg = wg_excl + before;
    const BlobOff bo = lmc_blob_off((u32)a.P, (u32)s.T, (u32)a.G);
    u8* const blob = a.blobs + (long long)chunk_i * a.blob_stride;
    u8* const out = blob + bo.streams + beg;
    counts_open_stream(s, cs, out, hist, lane);
    const u32 exact = cs.head + counts_code_stream<true>(a, s, hist, wring, rtab_lds, lane, reinterpret_cast<u16*>(out + cs.head));
    const u32 padded = (exact + 15u) & ~15u;
    if (alloc > padded) zero_fill16(out + padded, alloc - padded, lane);
    if (lane == 0) {
      u32* d = reinterpret_cast<u32*>(blob + bo.gdir) + 2 * pg_i;
      d[0] = beg;
      d[1] = beg + exact;
      if (exact > alloc) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);  // the bound is a theorem: never
    }
    if ((int)pg_i == n - 1) {  // the chunk's last stream knows the chunk's size: header, static sections, size word
      write_blob_static(blob, bo, a, (u32)s.T, wg_excl + wg_total, lane);
      if (lane == 0) a.sizes[chunk_i] = bo.streams + wg_excl + wg_total;
    }
    return;
  } else if constexpr (ENCODE && QUADSYM) {
    // chunks of 2 .. 256 tokens are coded on their symbol counts (LMC_MODEL_COUNTS), every other length on the 16-bit CDF
    const int chunk_of = (int)chunk_i;
    const bool counts_model =
        lmc_model_for_dev((u32)min(a.chunk_tokens, a.tok_end - (a.tok_begin + chunk_of * a.chunk_tokens))) == LMC_MODEL_COUNTS;  // wave-uniform
    if (counts_model) counts_stream();
    else {
      encode_group_stream<QUADSYM, ENCODE>(a, gid, hist, wring, lane, t);
      alloc = (t.exact + 15u) & ~15u;
    }
  } else {
    encode_group_stream<QUADSYM, ENCODE>(a, gid, hist, wring, lane, t);
    alloc = (t.exact + 15u) & ~15u;
  }
  if (!ENCODE) return;
  // ---- placement: where does this stream go? ------------------------------------------------------------
  const int n = a.P * a.G;
  const int chunk = t.chunk;
  unsigned long long* agg = a.agg + (long long)chunk * n;
  if (n % NW == 0) {
    // The waves of a workgroup hold consecutive streams of one chunk: they add their allocations up in LDS and
    // ONE wave runs the look-back over workgroup-level granules -- 1/NW of the granules, and of the
    // walk when a whole chunk finishes at once and nobody has an inclusive prefix yet.
    __shared__ u32 wg_alloc[NW];
    __shared__ u32 wg_excl;
    if (lane == 0) wg_alloc[wave] = alloc;
    __syncthreads();
    u32 before = 0, wg_total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      before += w < wave ? wg_alloc[w] : 0u;
      wg_total += wg_alloc[w];
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
    place_stream(a, t, wg_excl + before, alloc, wg_excl + wg_total, hist, lane);
  } else {
    // streams of a chunk do not fill whole workgroups: every wave publishes and looks back for itself
    if (lane == 0 && t.pg > 0) agg_store(agg + t.pg, AGG_A, alloc);
    const u32 excl = lookback_exclusive(agg, t.pg, lane, a.status);
    if (lane == 0) agg_store(agg + t.pg, AGG_P, excl + alloc);
    place_stream(a, t, excl, alloc, excl + alloc, hist, lane);
  }
}
