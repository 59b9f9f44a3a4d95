// k_bits.h -- a 64 x 64 bit-matrix transpose across a wave64: lane l holds row l (bit k of {lo, hi} = column k); on
// return lane k holds column k (bit l = what lane l had at bit k).  The head of a group stream (k_head.h) is such a
// matrix per 64 planes: a lane's counts, concatenated, are a row; a plane (one bit of one symbol's count over the 64
// lanes) is a column.
//
// Six block-swap stages, s = 1, 2, 4, 8, 16, 32: lanes l and l ^ s (l & s == 0) exchange the bits {row l, columns with
// k & s set} and {row l + s, columns with k & s clear}.  The partner's dword comes through ds_swizzle (the LDS
// crossbar, no LDS memory and -- what matters in kernels bound by vector issue -- no VALU slot); s = 1, 2, 4 rotate it
// by s and merge under a mask (v_alignbit + v_bfi), s = 8, 16 are byte permutes (v_perm), s = 32 is one
// v_permlane32_swap of the two dwords.  ~37 vector instructions per 64 planes, where one ballot + two v_writelane per
// plane cost 192.
#pragma once
#include "lmc_device.h"

template <int S>
__device__ __forceinline__ u32 lane_xor_dword(u32 v) {  // v of lane ^ S, S < 32
  return (u32)__builtin_amdgcn_ds_swizzle((int)v, (S << 10) | 0x1f);
}

__device__ __forceinline__ void transpose64(u32& lo, u32& hi, int lane) {
  static_for<3>([&](auto itag) {
    constexpr int i = decltype(itag)::value, s = 1 << i;
    constexpr u32 P = s == 1 ? 0xaaaaaaaau : s == 2 ? 0xccccccccu : 0xf0f0f0f0u;  // the columns with k & s set
    const u32 t = (u32)__builtin_amdgcn_sbfe(lane, i, 1);  // 0 in the pair's lower lane, ~0 in its upper one
    const u32 M = P ^ t;                          // where the partner's bits go
    const u32 r = (t & (2u * s)) - (u32)s;        // rotate right by: 32 - s below (= left by s), s above
    const u32 pl = lane_xor_dword<s>(lo), ph = lane_xor_dword<s>(hi);
    lo = (M & __builtin_amdgcn_alignbit(pl, pl, r)) | (~M & lo);  // one v_bfi_b32
    hi = (M & __builtin_amdgcn_alignbit(ph, ph, r)) | (~M & hi);
  });
  {
    const u32 sel = (lane & 8) ? 0x03070105u : 0x06020400u;
    const u32 pl = lane_xor_dword<8>(lo), ph = lane_xor_dword<8>(hi);
    lo = __builtin_amdgcn_perm(pl, lo, sel);
    hi = __builtin_amdgcn_perm(ph, hi, sel);
  }
  {
    const u32 sel = (lane & 16) ? 0x03020706u : 0x05040100u;
    const u32 pl = lane_xor_dword<16>(lo), ph = lane_xor_dword<16>(hi);
    lo = __builtin_amdgcn_perm(pl, lo, sel);
    hi = __builtin_amdgcn_perm(ph, hi, sel);
  }
  // s = 32: hi of lanes 0 .. 31 <-> lo of lanes 32 .. 63 (v_permlane32_swap_b32 x, y: lanes 32 .. 63 of x <-> lanes
  // 0 .. 31 of y)
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
