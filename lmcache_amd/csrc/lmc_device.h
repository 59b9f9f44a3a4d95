// lmc_device.h -- device-side helpers shared by the gfx950 kernels.
// CDNA4 only: wave64, no portability shims.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lmc_format.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#include "../../include/lmc_hip.h"

#define LMC_WAVE 64

// Device status bits (include/lmc_hip.h).
#define LMC_ST_STREAM_OVERFLOW LMC_STATUS_STREAM_OVERFLOW
#define LMC_ST_BAD_HEADER LMC_STATUS_BAD_HEADER
#define LMC_ST_BAD_STREAM LMC_STATUS_BAD_STREAM
#define LMC_ST_LOOKBACK_TIMEOUT LMC_STATUS_LOOKBACK_TIMEOUT
#define LMC_ST_BAD_SCALES LMC_STATUS_BAD_SCALES

// Device copy of lmc_kv_layout (include/lmc_hip.h), strides in elements.
struct KvAddr {
  const u16* base;
  const u16* const* plane_ptrs;
  const long long* slot_mapping;
  long long stride_layer, stride_kv, stride_token, stride_head, stride_block;
  int block_size;
  int L, H, D, dtype;
};

struct BinsArg {
  u8 b[LMC_MAX_PLANES];
};

__device__ __forceinline__ const u16* lmc_plane_base(const KvAddr& a, int p) {
  int kv = p >= a.L ? 1 : 0, l = p - kv * a.L;  // planes are K of every layer, then V (P = 2L)
  if (a.plane_ptrs) return a.plane_ptrs[2 * l + kv];
  return a.base + (long long)l * a.stride_layer + (long long)kv * a.stride_kv;
}

__device__ __forceinline__ long long lmc_tok_off(const KvAddr& a, int t) {
  if (a.slot_mapping) {
    u32 s = (u32)a.slot_mapping[t];
    u32 b = s / (u32)a.block_size;
    u32 w = s - b * (u32)a.block_size;
    return (long long)b * a.stride_block + (long long)w * a.stride_token;
  }
  return (long long)t * a.stride_token;
}

// Device twins of lmc_format.h's lmc_model_for / lmc_counts_scale_magic (ceil(2^32 / T) = floor((2^32 - 1) / T) + 1 for
// every T >= 2: a power of two divides 2^32, any other T does not).
__device__ __forceinline__ u32 lmc_model_for_dev(u32 T) {
  return (T >= LMC_COUNTS_T_MIN && T <= LMC_COUNTS_T) ? LMC_MODEL_COUNTS : LMC_MODEL_CDF16;
}
__device__ __forceinline__ bool lmc_model_valid_dev(u32 model, u32 T) {  // lmc_model_valid
  return model == LMC_MODEL_CDF16 || (model == LMC_MODEL_COUNTS && T >= LMC_COUNTS_T_MIN && T <= LMC_COUNTS_T);
}
__device__ __forceinline__ u32 lmc_counts_scale_magic_dev(u32 T) { return 0xffffffffu / T + 1u; }

// Workspace symbol format of a plane: symbols are 0 .. bins - 2, so planes with bins <= 17 pack two per byte
// (k_quantize.h writes, k_encode.h reads).
__device__ __forceinline__ bool lmc_sym_nibbles(int bins) { return bins <= 17; }

// Section offsets of a chunk blob with T tokens (mirror of lmc_blob_layout).
struct BlobOff {
  u32 bins, scales, scsum, gdir, streams;
};
__device__ __forceinline__ BlobOff lmc_blob_off(u32 P, u32 T, u32 G) {
  BlobOff o;
  o.bins = LMC_HEADER_BYTES;
  o.scales = o.bins + ((P + 15u) & ~15u);
  o.scsum = o.scales + ((2u * P * T + 15u) & ~15u);
  o.gdir = o.scsum + ((4u * P + 15u) & ~15u);
  o.streams = o.gdir + ((8u * P * G + 15u) & ~15u);  // directory entries are {beg, end} pairs
  return o;
}

// e / R and e % R for small uniform R (2..30) and e < 2^15: exact through one float multiply
// ((e + 0.5) / R is never within 1/60 of an integer; checked exhaustively on the host).
__device__ __forceinline__ void divmod_small(u32 e, u32 R, float rcpR, u32& q, u32& r) {
  q = (u32)(((float)e + 0.5f) * rcpR);
  r = e - q * R;
}

// Pointers that come out of a device pointer table are "generic" to the compiler and would be
// accessed with flat_* instructions (which tick both vmcnt and lgkmcnt); KV and blobs always live
// in global memory, so say so.
#define LMC_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld_global_u4(const u16* p) {
  const u32x4_t v = *reinterpret_cast<const LMC_GLOBAL u32x4_t*>((const LMC_GLOBAL u16*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
// the same load with the non-temporal hint: streamed-once data (the raw KV) should not push the workspace out of the caches
__device__ __forceinline__ uint4 ld_global_u4_nt(const u16* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const LMC_GLOBAL u32x4_t*>((const LMC_GLOBAL u16*)p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_global_u4(u16* p, uint4 v) {
  u32x4_t t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *reinterpret_cast<LMC_GLOBAL u32x4_t*>((LMC_GLOBAL u16*)p) = t;
}
__device__ __forceinline__ void st_global_u16(u16* p, u16 v) { *((LMC_GLOBAL u16*)p) = v; }

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// max of two packed u16 pairs (v_pk_max_u16)
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
  us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
  us2 z = __builtin_elementwise_max(x, y);
  return __builtin_bit_cast(u32, z);
}

template <int DT>
__device__ __forceinline__ float h_lo(u32 w) {  // low 16-bit element of a packed pair -> fp32
  if (DT == LMC_DTYPE_BF16) return __uint_as_float(w << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
}
template <int DT>
__device__ __forceinline__ float h_hi(u32 w) {
  if (DT == LMC_DTYPE_BF16) return __uint_as_float(w & 0xffff0000u);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
}
__device__ __forceinline__ float h2f_rt(u32 bits, int dtype) {
  if (dtype == LMC_DTYPE_BF16) return __uint_as_float(bits << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
}

// fp32 -> bf16 bits, round-to-nearest-even; NaN -> 0x7FC0 (c10::BFloat16).
__device__ __forceinline__ u32 f2bf16(float f) {
  u32 u = __float_as_uint(f);
  u32 r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  return (f != f) ? 0x7fc0u : r;
}
__device__ __forceinline__ u32 f2fp16(float f) {  // v_cvt_f16_f32, RNE
  return (u32)__builtin_bit_cast(unsigned short, (_Float16)f);
}

// factor = MAX / max of a quantiser row, IEEE fp32 (the reference divides: cachegen_encoder.py:40-61).  The
// compiler's division is 14 instructions (v_div_scale x 2, v_rcp, five fma / mul, v_div_fmas, v_div_fixup),
// executed by every lane for a wave-uniform value.  Rows whose max lies well inside the fp32 normal range take
//   y = rcp(max); q0 = MAX * y; r = fma(-max, q0, MAX); q = fma(r, y, q0)                      (4 instructions)
// which tools/probes/row_div.hip checks exhaustively ON THE GPU (v_rcp_f32 is a hardware approximation) against
// the division over every 16-bit max x MAX 1 .. 15: 0 mismatches inside row_div_in_range for bf16 and fp16
// (profiles/r03_row_div.log; outside it -- bf16 maxes that are fp32 denormals or whose reciprocal is -- 4014 of
// 491 520 differ, which is why the range test exists).  The test is on the row max, wave-uniform: a scalar branch;
// zero / inf / NaN maxes fail it and keep the division.
#ifndef LMC_SHORT_ROW_DIV
#define LMC_SHORT_ROW_DIV 1
#endif
__device__ __forceinline__ bool row_div_in_range(u32 max_bits, int dtype) {
  if (dtype == LMC_DTYPE_BF16) {
    const u32 e = (max_bits >> 7) & 0xffu;  // bf16 carries fp32's exponent field
    return e >= 27u && e <= 227u;           // 2^-100 .. 2^100
  }
  return max_bits != 0u && max_bits < 0x7c00u;  // fp16: any finite non-zero max (1 / max is a normal fp32)
}
__device__ __forceinline__ float row_div_short(float maxf, float sf) {
  const float y = __builtin_amdgcn_rcpf(sf);
  const float q0 = maxf * y;
  const float r = __builtin_fmaf(-sf, q0, maxf);
  return __builtin_fmaf(r, y, q0);
}

// a pointer every lane holds -> an SGPR pair (so that loads through it take the scalar-base + 32-bit lane-offset form)
__device__ __forceinline__ u64 uniform_ptr64(const void* p) {
  const u64 v = (u64)p;
  const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
  return ((u64)hi << 32) | (u64)lo;
}

// Order this wave's LDS traffic (cross-lane hand-off inside one wave): a
// compiler + hardware fence at wavefront scope; no workgroup barrier needed.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// max over the 64 lanes of a wave for TWO values at once, on the VALU's data-parallel-primitive path (no LDS round
// trips: __shfl_xor is ds_bpermute_b32 + s_waitcnt, six of them in a row on the quantise stage's critical path):
// quad swaps, half-row and row mirrors leave every row of 16 lanes with its maximum, row_bcast15 / row_bcast31 carry
// it into the rows above, lane 63 holds the wave's maximum.  The two chains are interleaved so that each DPP read
// has its two wait states behind the write of its register (one of them is the other chain's instruction).  The
// results are wave-uniform (SGPRs).
__device__ __forceinline__ void wave_max2_u32(u32& a, u32& b) {
  u32 sa, sb;
  asm("s_nop 1\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %[a], %[a], %[a] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_max_u32_dpp %[b], %[b], %[b] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_readlane_b32 %[sa], %[a], 63\n\t"
      "v_readlane_b32 %[sb], %[b], 63\n\t"
      "s_nop 4"  // whatever follows may read the two SGPRs at once (the compiler does not look inside the block)
      : [a] "+v"(a), [b] "+v"(b), [sa] "=s"(sa), [sb] "=s"(sb));
  a = sa;
  b = sb;
}

// bitwise OR over the 64 lanes for TWO values at once, the same DPP ladder; wave-uniform results
__device__ __forceinline__ void wave_or2_u32(u32& a, u32& b) {
  u32 sa, sb;
  asm("s_nop 1\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_or_b32_dpp %[b], %[b], %[b] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_readlane_b32 %[sa], %[a], 63\n\t"
      "v_readlane_b32 %[sb], %[b], 63\n\t"
      "s_nop 4"
      : [a] "+v"(a), [b] "+v"(b), [sa] "=s"(sa), [sb] "=s"(sb));
  a = sa;
  b = sb;
}

// ... for FOUR values at once, as a reduce-scatter: two half-swaps fold the four registers into one whose row r
// (16 lanes) holds value r OR-ed over the four rows -- v_permlane32_swap_b32 x, y exchanges lanes 32..63 of x with lanes
// 0..31 of y, so x | y afterwards is "a over the lane pair (l, l + 32)" in the lower half and "c ..." in the upper;
// v_permlane16_swap_b32 does the same with odd and even rows -- then the four DPP steps inside the rows, and one
// v_readlane per value.  14 VALU instructions where four separate ladders take 56.
__device__ __forceinline__ void wave_or4_u32(u32& a, u32& b, u32& c, u32& d) {
  u32 sa, sb, sc, sd;
  asm("s_nop 1\n\t"
      "v_permlane32_swap_b32 %[a], %[c]\n\t"
      "v_permlane32_swap_b32 %[b], %[d]\n\t"
      "s_nop 1\n\t"
      "v_or_b32_e32 %[a], %[a], %[c]\n\t"
      "v_or_b32_e32 %[b], %[b], %[d]\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %[a], %[b]\n\t"
      "s_nop 1\n\t"
      "v_or_b32_e32 %[a], %[a], %[b]\n\t"
      "s_nop 1\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_or_b32_dpp %[a], %[a], %[a] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %[sa], %[a], 0\n\t"
      "v_readlane_b32 %[sb], %[a], 16\n\t"
      "v_readlane_b32 %[sc], %[a], 32\n\t"
      "v_readlane_b32 %[sd], %[a], 48\n\t"
      "s_nop 4"
      : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [sa] "=s"(sa), [sb] "=s"(sb), [sc] "=s"(sc), [sd] "=s"(sd));
  a = sa; b = sb; c = sc; d = sd;
}

// sum over the 64 lanes on the same DPP ladder (every lane of a quad holds the quad's sum, a half-row mirror adds the
// other quad of the half, ...; row_bcast adds the rows below into rows 1 and 3, then into 2 and 3): lane 63 holds the
// total.  Wave-uniform result.
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
  u32 sv;
  asm("s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %[v], %[v], %[v] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %[sv], %[v], 63\n\t"
      "s_nop 4"
      : [v] "+v"(v), [sv] "=s"(sv));
  return sv;
}

// Checksum of a plane's T scales (lmc_format.h: scsum), by one wave.
__device__ __forceinline__ u32 scale_checksum(const u16* scl, u32 T, int lane) {
  u32 acc = 0;
  for (u32 t = (u32)lane; t < T; t += 64) acc += (t + 1u) * ((u32)scl[t] + 1u);  // device twin of lmc_scale_checksum_term
  return wave_sum_u32(acc);
}

__device__ __forceinline__ u32 lane_rank(u64 mask) {  // # set bits of mask below this lane
  return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// RNE(v / T) for wave-uniform T; magic = floor(2^32 / T) (0xffffffff for T == 1).
// v = n * 65504 with n <= T < 65536, so v < 2^32.
__device__ __forceinline__ u32 rne_div_u32(u32 v, u32 T, u32 magic) {
  u32 q = __umulhi(v, magic);
  u32 r = v - q * T;
  if (r >= T) { q++; r -= T; }
  if (r >= T) { q++; r -= T; }
  u32 r2 = 2u * r;
  q += (r2 > T || (r2 == T && (q & 1u))) ? 1u : 0u;
  return q;
}

// Exact (x / f, x % f) for x < f * 2^16, 1 <= f < 2^16.
// qe = trunc(float(x) * rcp(f) * (1 - 2^-21)): the four roundings involved
// (cvt 2^-24, v_rcp_f32 1 ulp = 2^-23, two multiplies 2^-24 each) add up to
// < 2^-21.6 relative, so the biased estimate never exceeds the true quotient
// and is short of it by < 2^16 * 2^-20.3 < 1: qe is q or q-1, one branch-free
// correction (re >= f) finishes it.
__device__ __forceinline__ void divmod_est(u32 x, u32 f, u32& q, u32& r) {
  const float rf = __builtin_amdgcn_rcpf((float)f) * 0.99999952316284f;  // 1 - 2^-21
  const u32 qe = (u32)((float)x * rf);
  const u32 re = x - __umul24(qe, f);
  const bool fix = re >= f;
  q = qe + (fix ? 1u : 0u);
  r = fix ? re - f : re;
}

// The 33-entry CDF column of this lane's channel from its symbol counts (hreg[k] = count of symbol 2k |
// count of symbol 2k+1 << 16), written to LDS as tab[entry][lane] u16 (entry 32 is 65536 stored as 0):
//   cdf[i] = RNE(N_i * 65504 / T) + i,  N_i = number of symbols < i   (cachegen_encoder.py:95-126)
// in exact integer arithmetic.  Symbols are 0 .. nsym - 1, so N_i = T and cdf[i] = 65504 + i from entry nsym
// on whatever the data: only entries 1 .. nsym - 1 are divided, by a shift when T is a power of two (the
// usual 256-token chunk).  Shared by the encoder (counts from its histogram) and the decoder (counts from
// the blob).
__device__ __forceinline__ void cdf_column_to_lds(const u32 (&hreg)[16], u32 T, u32 nsym, u16* tab, int lane) {
  const u32 magic = (T == 1u) ? 0xffffffffu : (u32)(0x100000000ull / T);
  const bool pow2 = (T & (T - 1u)) == 0u;
  const u32 sh = 31u - (u32)__builtin_clz(T);  // log2(T) when pow2
  const u32 half_m1 = sh ? (1u << (sh - 1u)) - 1u : 0u;
  u32 n = 0;
#pragma unroll
  for (int i = 0; i <= 32; i++) {
    u32 ci = LMC_CDF_SCALE + (u32)i;
    if (i == 0) {
      ci = 0;
    } else if ((u32)i < nsym) {
      const u32 v = n * LMC_CDF_SCALE;
      if (pow2) ci = (sh ? (v + half_m1 + ((v >> sh) & 1u)) >> sh : v) + (u32)i;  // round half to even
      else ci = rne_div_u32(v, T, magic) + (u32)i;
    }
    tab[i * 64 + lane] = (u16)ci;
    if (i < 32) n += (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
  }
}

// The same column for planes with at most 16 symbols, in the decoder's packed form: 16 entries of 32 bits,
// entry i = cdf[i] << 16 | (cdf[i + 1] - cdf[i]) (start and frequency of symbol i; cdf[16] <= 65520), stored as
// tab32[i / 4][lane][i % 4]: a lane's four entries of a quarter are one aligned 16-byte read, bank-conflict free.
__device__ __forceinline__ void cdf_column_to_lds_wide(const u32 (&hreg)[16], u32 T, u32 nsym, u32* tab32, int lane) {
  const u32 magic = (T == 1u) ? 0xffffffffu : (u32)(0x100000000ull / T);
  const bool pow2 = (T & (T - 1u)) == 0u;
  const u32 sh = 31u - (u32)__builtin_clz(T);
  const u32 half_m1 = sh ? (1u << (sh - 1u)) - 1u : 0u;
  u32 n = hreg[0] & 0xffffu, prev = 0;
#pragma unroll
  for (int i = 1; i <= 16; i++) {
    u32 ci = LMC_CDF_SCALE + (u32)i;
    if ((u32)i < nsym) {
      const u32 v = n * LMC_CDF_SCALE;
      if (pow2) ci = (sh ? (v + half_m1 + ((v >> sh) & 1u)) >> sh : v) + (u32)i;
      else ci = rne_div_u32(v, T, magic) + (u32)i;
    }
    tab32[((i - 1) >> 2) * 256 + lane * 4 + ((i - 1) & 3)] = (prev << 16) | (ci - prev);
    prev = ci;
    if (i < 16) n += (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
  }
}

// One rANS state update x' = ((x / f) << 16) + (x % f) + st for x < f * 2^16, 1 <= f < 2^16,
// c = 2^16 - f.  With q = x / f:  x' = x + q * c + st, so only the quotient is needed.
// qe = trunc(fma(float(x), rcp(f), -0.05)): float(x) is off by <= 2^-24 relative, v_rcp_f32 by
// <= 1 ulp (2^-23), the fma rounds once (<= 2^-9 absolute below 2^16); with x / f < 2^16 the sum
// is < 0.014 (< 0.03 even at 4 ulp), so the biased estimate never exceeds x / f and is short of it
// by < 0.1: qe is q or q - 1, and q = qe + (x >= (qe + 1) * f).  (qe + 1) * f <= x + f < 2^32.
__device__ __forceinline__ u32 rans_put(u32 x, u32 f, u32 c, u32 st) {
  const float qf = __builtin_fmaf((float)x, __builtin_amdgcn_rcpf((float)f), -0.05f);
  const u32 qe = (u32)qf;
  const u32 q = qe + (x >= __umul24(qe, f) + f ? 1u : 0u);
  u32 t;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(q), "v"(c), "v"(x));  // q * c + x in one op
  return t + st;
}

template <int I>
struct IntTag { static constexpr int value = I; };  // compile-time integer passed as a value
template <bool B>
struct BoolTag { static constexpr bool value = B; };
// Compile-time loop: f(IntTag<0>{}), f(IntTag<1>{}), ...
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntTag<I>{});
    static_for<N, I + 1>(f);
  }
}

// e / 33 for e < 8192 (CDF rows are 33 entries)
__device__ __forceinline__ u32 div33(u32 e) { return (e * 1986u) >> 16; }
