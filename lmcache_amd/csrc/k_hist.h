// k_hist.h -- the plane histogram a fused encode takes WHILE it quantises (round 6; k_fused.h has the why and the layout):
// counters of a whole work item (one plane of <= 1024 channels, or up to 8 narrower planes) in the workgroup's eight
// idle 4 KiB table slices, at LDS address 0.
//   <= 16 symbols  u16 counters; block b = it * 4 + e / 2 (4 KiB) is [16 symbols][64 lanes] dwords, half e & 1
//   more symbols   u8 counters;  block b = it * 2 + e / 4 (8 KiB) is [32 symbols][64 lanes] dwords, byte e & 3
// (it, lane) = the VIRTUAL quantising lane of the channel: element e of channel run `it` of lane `lane` of a wave that
// quantises 1024 channels at once; a narrow plane pj of an item maps its lane group onto lanes (pj * GL + sl) & 63 of
// run (pj * GL + sl) >> 6, so every plane of the item owns its own columns and may use its own counter format.
#pragma once
#include "lmc_device.h"

#define PLANE_HIST_DWORDS 8192

// one histogram add of the plane: counter row `sym` of the lane's column, `field` = the symbol moved to bits 8 .. 12
template <int OFF>
__device__ __forceinline__ void plane_hist_add(u32 ad, u32 val) {
  typedef __attribute__((address_space(3))) u32* lds_u32w;
  __hip_atomic_fetch_add((lds_u32w)(size_t)(ad + (u32)OFF), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS byte address of a symbol's counter row in the lane's column: `ad` holds 4 * lane in byte 0 (the histogram starts
// at LDS address 0 -- checked -- and rows are 256 B), and ONE SDWA instruction writes the symbol into byte 1 and keeps
// the rest (dst_unused:UNUSED_PRESERVE): the low nibble of byte K of w (v_and 15), its high nibble (v_lshrrev 4), or
// the whole byte (a byte plane's symbol is < 32).  (Shift + v_and_or_b32 takes two.)
#define LMC_HIST_ROW_OP(NAME, OPSTR)                                                                                     \
  template <int K>                                                                                                      \
  __device__ __forceinline__ void NAME(u32& ad, u32 w) {                                                                \
    if constexpr (K == 0) asm(OPSTR " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0" : "+v"(ad) : "v"(w)); \
    else if constexpr (K == 1) asm(OPSTR " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1" : "+v"(ad) : "v"(w)); \
    else if constexpr (K == 2) asm(OPSTR " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(ad) : "v"(w)); \
    else asm(OPSTR " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3" : "+v"(ad) : "v"(w)); \
  }
LMC_HIST_ROW_OP(hist_row_lo, "v_and_b32_sdwa %0, 15, %1")      // byte 1 = byte K of w & 15
LMC_HIST_ROW_OP(hist_row_hi, "v_lshrrev_b32_sdwa %0, 4, %1")   // byte 1 = byte K of w >> 4
LMC_HIST_ROW_OP(hist_row_byte, "v_or_b32_sdwa %0, 0, %1")      // byte 1 = byte K of w


// The symbols of one workspace dword `od` -- element E of run IT -- into the histogram.  NIB: eight nibbles, token row =
// byte (row & 3), high nibble from row 4 on; else four bytes, token row 4 * HQ + byte.  counted(row) says whether the
// token is counted (a token of the chunk; not token 0 of a byte plane); ad = two address registers holding 4 * lane in
// byte 0, used in turn.
template <bool NIB, int IT, int E, int HQ, class Pred>
__device__ __forceinline__ void plane_hist_dword(u32 od, u32 (&ad)[2], Pred&& counted) {
  if constexpr (NIB) {
    constexpr int OFF = (IT * 4 + E / 2) * 4096;
    const u32 val = (E & 1) ? 0x10000u : 1u;
    static_for<8>([&](auto ktag) {
      constexpr int row = decltype(ktag)::value;
      if (counted(row)) {
        if constexpr (row < 4) hist_row_lo<row & 3>(ad[row & 1], od);
        else hist_row_hi<row & 3>(ad[row & 1], od);
        plane_hist_add<OFF>(ad[row & 1], val);
      }
    });
  } else {
    constexpr int OFF = (IT * 2 + E / 4) * 8192;
    const u32 val = 1u << (8 * (E & 3));
    static_for<4>([&](auto ktag) {
      constexpr int k = decltype(ktag)::value;
      if (counted(4 * HQ + k)) {
        hist_row_byte<k>(ad[k & 1], od);
        plane_hist_add<OFF>(ad[k & 1], val);
      }
    });
  }
}
