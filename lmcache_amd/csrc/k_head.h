// k_head.h -- the head of a group stream (include/lmc_format.h, v6): the symbol counts of the stream's 64 channels,
// bit-sliced.  Takes the place of the container's [2L, C, 33] `cdf` tensor (cachegen_basics.py:109-142,
// cachegen_encoder.py:287-290): a channel's CDF is a function of its counts (lmc_device.h: cdf_column_to_lds).
//
//   widths u8[R8] | planes u64[W] | zeros to 16 B        R8 = R rounded up to 8, W = sum of the widths
//
// A plane is one bit of one symbol's stored count across the 64 lanes: a COLUMN of the 64 x 64 bit matrix whose rows
// are the lanes' counts strung together -- 64 planes at a time are made from, and turned back into, the lanes' rows by
// one transpose64 (k_bits.h); widths[i] is wave-uniform (an OR over the lanes), so every loop below runs on scalar
// trip counts and every field sits at a wave-uniform bit position.
#pragma once
#include "lmc_device.h"
#include "k_bits.h"

// v_writelane_b32: lane SEL of `old` <- the wave-uniform `val`.
template <int SEL>
__device__ __forceinline__ int writelane_const(int val, int old) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane(val)), "n"(SEL));
  return old;
}
// Significant bits of the wave-uniform value v (0 for 0).
__device__ __forceinline__ u32 bit_width_u32(u32 v) { return v ? 32u - (u32)__builtin_clz(v) : 0u; }

// The widths of the symbols' fields from the lanes' stored counts, packed FB bits per field in pk[0 .. NREG) (symbol i
// = field i % (32 / FB) of pk[i / (32 / FB)]): wor[k] = OR over the lanes of pk[k] (wave-uniform), from which
// field_width() reads a symbol's width.
template <int NREG>
__device__ __forceinline__ void head_or_counts(const u32 (&pk)[NREG], u32 (&wor)[NREG]) {
  static_assert(NREG % 4 == 0, "four registers per reduction");
#pragma unroll
  for (int k = 0; k < NREG; k += 4) {
    u32 a = pk[k], b = pk[k + 1], c = pk[k + 2], d = pk[k + 3];
    wave_or4_u32(a, b, c, d);
    wor[k] = a;
    wor[k + 1] = b;
    wor[k + 2] = c;
    wor[k + 3] = d;
  }
}
template <int FB, int NREG>
__device__ __forceinline__ u32 field_width(const u32 (&wor)[NREG], int i) {
  constexpr int PER = 32 / FB;
  return bit_width_u32((wor[i / PER] >> (FB * (i % PER))) & ((1u << FB) - 1u));
}
// Bytes of the head of a plane with R symbols whose counts OR to `wor` (lmc_head_bytes).
template <int FB, int NREG>
__device__ __forceinline__ u32 head_bytes_of(const u32 (&wor)[NREG], u32 R) {
  u32 W = 0;
  static_for<NREG * (32 / FB)>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    if ((u32)i < R) W += field_width<FB, NREG>(wor, i);
  });
  return (((R + 7u) & ~7u) + 8u * W + 15u) & ~15u;
}

// Write the head to `out` (16-byte aligned, wave-uniform); returns its size.  pk: this lane's stored counts (0 for a
// lane without a channel), wor: their OR over the lanes (head_or_counts).  `stage`: 512 bytes of this wave's LDS per 64
// planes (8 * W bytes; the wave's idle table slice).
// A lane strings its counts together, most significant bit first, 64 bits at a time -- a ROW of the 64 x 64 bit matrix
// whose COLUMNS are the planes -- and parks each finished word in its own LDS slot (a compile-time loop over the
// symbols: pk and wor stay in registers; two or three instructions per symbol); then, per 64 planes, one transpose64
// (k_bits.h) turns rows into planes and one coalesced 8-byte store per lane writes them.  (A ballot and two
// v_writelane per plane, the first form of this function, cost three instructions per PLANE.)
template <int FB, int NREG>
__device__ __forceinline__ u32 head_write(u8* out_v, const u32 (&pk)[NREG], const u32 (&wor)[NREG], u32 R, u32* stage, int lane) {
  constexpr int PER = 32 / FB, NSYM = NREG * PER;
  typedef __attribute__((address_space(3))) u32x2_t* lds_u64w;
  u8* const out = reinterpret_cast<u8*>(uniform_ptr64(out_v));
  const u32 R8 = (R + 7u) & ~7u;
  LMC_GLOBAL u32x2_t* const planes = (LMC_GLOBAL u32x2_t*)(out + R8);
  const lds_u64w slot = (lds_u64w)reinterpret_cast<u32x2_t*>(stage) + lane;  // word b of this lane: slot[64 b]
  int wv = 0;    // lane i: widths[i]
  u64 acc = 0;   // the lane's row of the current block: nb bits, right-aligned
  u32 nb = 0;    // (wave-uniform, like every width)
  u32 nblk = 0;  // finished words
  static_for<NSYM>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    const u32 w = (u32)i < R ? field_width<FB, NREG>(wor, i) : 0u;  // uniform
    if (w != 0u) {  // (uniform; a symbol that does not occur in the group costs nothing: widths starts at 0)
      wv = writelane_const<i>((int)w, wv);
      const u32 c = (pk[i / PER] >> (FB * (i % PER))) & ((1u << FB) - 1u);  // < 2^w
      if (nb + w < 64u) {
        acc = (acc << w) | c;
        nb += w;
      } else {  // the word is complete with the top 64 - nb bits of the field
        const u32 t = 64u - nb, rest = w - t;
        const u64 full = (acc << t) | (u64)(c >> rest);
        slot[64u * nblk] = u32x2_t{(u32)full, (u32)(full >> 32)};
        nblk++;
        acc = (u64)(c & ((1u << rest) - 1u));
        nb = rest;
      }
    }
  });
  const u32 W = 64u * nblk + nb;
  if (nb != 0u) {  // the last, partial word, left-aligned: plane p of the block is bit 63 - p whatever the fill
    const u64 full = acc << (64u - nb);
    slot[64u * nblk] = u32x2_t{(u32)full, (u32)(full >> 32)};
  }
  for (u32 b = 0; 64u * b < W; b++) {
    const u32x2_t v = slot[64u * b];
    u32 lo = v.x, hi = v.y;
    transpose64(lo, hi, lane);  // lane k: column k = plane 63 - k of the block
    const u32 p = 64u * b + 63u - (u32)lane;
    if (p < W) planes[p] = u32x2_t{lo, hi};
  }
  if ((u32)lane < R8) ((LMC_GLOBAL u8*)out)[lane] = (u8)wv;  // widths[R .. R8) = 0: wv starts at 0
  const u32 raw = R8 + 8u * W;
  if ((raw & 8u) && lane == 0) planes[W] = u32x2_t{0u, 0u};  // zeros to 16 bytes
  return (raw + 15u) & ~15u;
}

// Read a head back: `head` (wave-uniform, 16-byte aligned) -> this lane's stored counts cv[0 .. NSYM), NSYM = 32 or 16
// (symbols >= R read 0).  `stage` = 512 bytes of this wave's LDS per 64 planes (at most 16 * NSYM planes: 4 KiB for
// NSYM = 32); `limit` = bytes the head may take.  Returns the head's size, 0 if it is malformed (a width above 16, a
// head longer than `limit`).
// head_write backwards: per 64 planes, lane k loads plane 63 - k of the block (one coalesced 8-byte load), transpose64
// turns the planes into the lanes' rows, each lane parks its row in its own LDS slot; then a compile-time loop over
// the symbols cuts the counts out of the rows at wave-uniform bit positions (a 64-bit shift and a mask per symbol).
template <int NSYM>
__device__ __forceinline__ u32 head_read(const u8* head_v, u32 limit, u32 R, u32* stage, u32 (&cv)[NSYM], int lane) {
  typedef __attribute__((address_space(3))) u32x2_t* lds_u64w;
  const u8* const head = reinterpret_cast<const u8*>(uniform_ptr64(head_v));
  const u32 R8 = (R + 7u) & ~7u;
  // (loads below are unconditional, out-of-range lanes read the head's first bytes -- a stream is at least 256 bytes long --
  // and drop them: a load under a lane condition joins with a constant, and the register's other write waits for vmcnt(0))
  const u32 wv_raw = (u32)((const LMC_GLOBAL u8*)head)[lane];
  const u32 wv = ((u32)lane < R8 && R8 <= limit) ? wv_raw : 0u;
  // (round 6) the first PRE blocks of 64 planes are requested together with the widths -- W, the number of planes, is
  // only known from the widths, but where plane p lies is not a function of them, and every plane below W lies inside
  // `limit` (checked below): one round trip to memory instead of two for every head of up to 64 PRE planes
  constexpr int PRE = 3;
  const LMC_GLOBAL u32x2_t* const planes = (const LMC_GLOBAL u32x2_t*)(head + R8);
  u32x2_t pre[PRE];
#pragma unroll
  for (int b = 0; b < PRE; b++) {
    const u32 p = 64u * (u32)b + 63u - (u32)lane;
    const bool in = (unsigned long long)R8 + 8ull * p + 8ull <= (unsigned long long)limit;
    pre[b] = *(const LMC_GLOBAL u32x2_t*)(head + (in ? R8 + 8u * p : 0u));
  }
  u32 W = 0;
  bool bad = R8 > limit || R > (u32)NSYM;
  u32 w[NSYM];
  static_for<NSYM>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    w[i] = (u32)i < R ? (u32)__builtin_amdgcn_readlane((int)wv, i) : 0u;
    bad |= w[i] > 16u;
    W += w[i];
  });
  const u32 hb = (R8 + 8u * W + 15u) & ~15u;
  if (bad || hb > limit || W > 16u * (u32)NSYM) {
#pragma unroll
    for (int i = 0; i < NSYM; i++) cv[i] = 0u;
    return 0u;
  }
  const lds_u64w slot = (lds_u64w)reinterpret_cast<u32x2_t*>(stage) + lane;  // row b of this lane: slot[64 b]
  wave_lds_fence();
#pragma unroll
  for (int b = 0; b < PRE; b++) {
    if (64u * (u32)b < W) {  // (wave-uniform)
      const u32 p = 64u * (u32)b + 63u - (u32)lane;
      u32 lo = p < W ? pre[b].x : 0u, hi = p < W ? pre[b].y : 0u;
      transpose64(lo, hi, lane);  // lane l: its bits of planes 64 b .. 64 b + 63, the first plane on top
      slot[64u * (u32)b] = u32x2_t{lo, hi};
    }
  }
  for (u32 b = PRE; 64u * b < W; b++) {
    const u32 p = 64u * b + 63u - (u32)lane;
    u32x2_t v = {0u, 0u};
    if (p < W) v = planes[p];
    u32 lo = v.x, hi = v.y;
    transpose64(lo, hi, lane);
    slot[64u * b] = u32x2_t{lo, hi};
  }
  u64 cur = 0;   // the row being cut up
  u32 pos = 64;  // bits of it already used (wave-uniform)
  u32 blk = 0;   // rows fetched
  auto next_row = [&]() {
    const u32x2_t v = slot[64u * blk];
    blk++;
    return ((u64)v.y << 32) | (u64)v.x;
  };
  static_for<NSYM>([&](auto itag) {
    constexpr int i = decltype(itag)::value;
    const u32 wi = w[i];
    u32 c = 0;
    if (wi != 0u) {  // (uniform)
      if (pos + wi <= 64u) {
        c = (u32)(cur >> (64u - pos - wi)) & ((1u << wi) - 1u);
        pos += wi;
      } else {  // the field runs over into the next row (or starts one)
        const u32 t = 64u - pos, rest = wi - t;  // t < 16 bits left here
        const u32 top = (u32)cur & ((1u << t) - 1u);
        cur = next_row();
        c = (top << rest) | (u32)(cur >> (64u - rest));
        pos = rest;
      }
    }
    cv[i] = c;
  });
  wave_lds_fence();  // the staging is dead
  return hb;
}
