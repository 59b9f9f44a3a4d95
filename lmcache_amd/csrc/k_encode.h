// k_encode.h -- per-(plane, 64-channel group) histogram -> CDF -> interleaved
// rANS encode, then stream compaction.
//
// Replaces torchac_cuda.calculate_cdf (call site cachegen_encoder.py:287-289,
// in-tree spec :95-126, :175-222), torchac_cuda.encode_fast_new (:255-260) and
// collect_bytes (:225-238).  Stream format: include/lmc_format.h.
//
// One wave = one group stream: lane = channel.  Everything a wave needs is
// wave-private (its LDS slice, its scratch slot), so there is no barrier in
// this kernel up to the final placement; ENC_WAVES waves share a workgroup to share the CU and one look-back.
//
//   pass 1  per-lane histogram in LDS, u16 counters [symbol][lane]; lanes 2i and
//           2i+1 share a dword and add 1 / 1 << 16 with ds_add_u32 (bank = lane / 2).
//           The counts open the stream, bit-sliced (k_head.h; lmc_format.h "head").
//   CDF     cdf[i] = RNE(n_i * 65504 / T) + i, exact integer arithmetic
//           (cdf_column_to_lds, shared with the decoder), kept in LDS as
//           tab[entry][lane] u16 (aliases the dead histogram: 4.2 KiB per wave)
//   pass 2  tokens T-1..0: renormalise -- under exec = emitting lanes (v_cmpx on the state's upper half): mbcnt
//           rank, ds_write_b16 of the low half into a 256-word LDS ring (+ 64 slots so that a step never wraps;
//           flushed 256 B at a time), x >>= 16 -- then x = (x/f << 16) + x%f + start, computed as
//           x + (x/f) * (2^16 - f) + start (rans_put)
//   tail    64 states, zero pad to 16 B; then the stream is moved to its final place in the blob
//           (single-pass prefix over the group lengths of the chunk, one look-back per workgroup)
// 4.75 KiB of LDS per wave -> 8 waves per SIMD.
#pragma once
#include "k_head.h"

struct EncodeArgs {
  // symbols
  const u32* sym4;      // QUADSYM: symbol workspace, TQ*C dwords per (chunk, plane), format per plane (k_quantize.h)
  const int8_t* sym8;   // !QUADSYM: [P][T][C]
  int tok_begin, tok_end, chunk_tokens, nchunks;
  int P, C, G, TQ;
  long long sym_stride;  // dwords between the workspace regions of consecutive (chunk, plane) pairs: >= TQ * C (lmc_api.hip pads it)
  // outputs
  u8* blobs;            // ENCODE: blob i at blobs + i*blob_stride
  long long blob_stride;
  u16* cdf_out;         // !ENCODE: [P][C][33]
  u8* scratch;          // [nchunks*P*G][cap] padded group streams
  u32 cap;              // bytes per scratch slot
  u32* status;
  BinsArg bins;         // ENCODE: bins per plane
  // in-kernel stream compaction (single-pass prefix over the allocations of a chunk's streams)
  unsigned long long* agg;  // [nchunks][P*G], zeroed before the launch: flag << 62 | value
  u32* sizes;               // [nchunks] total blob bytes
  int L, H, D, dtype;       // for the header
  // Work tickets.  The look-back of the in-kernel compaction waits for workgroups with LOWER work indices; taking
  // the work index from blockIdx assumes the hardware starts workgroups in index order, which HIP does not
  // promise.  Instead every workgroup draws a ticket from a device counter when it STARTS (one relaxed agent-scope
  // atomic add by one lane) and works on item `ticket - ticket_base`: whoever holds a lower index has started
  // already and never waits for a later one, whatever the dispatch order.  The counter is never reset: the host
  // knows how many tickets earlier launches drew (ticket_base); launches sharing a context run on ordered streams.
  u32* ticket;
  u32 ticket_base;
};

// The calling workgroup's work index (see EncodeArgs::ticket); ends with a workgroup barrier.
__device__ __forceinline__ u32 draw_ticket(u32* ticket, u32 base) {
  __shared__ u32 wg_ticket;
  if (threadIdx.x == 0) wg_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
  __syncthreads();
  return wg_ticket;
}

// 64-bit {flag, value} granules of the decoupled look-back: one naturally aligned 8-byte agent-scope
// store / load each, so the value and its flag can never be seen torn or out of order.
#define AGG_X 0ull  // not published yet
#define AGG_A 1ull  // value = this group's own padded length
#define AGG_P 2ull  // value = inclusive prefix up to and including this group
__device__ __forceinline__ void agg_store(unsigned long long* p, unsigned long long flag, u32 v) {
  __hip_atomic_store(p, (flag << 62) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long agg_load(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Header, bins and the zero pads between sections (what the oracle memsets): by one wave.
__device__ __forceinline__ void write_blob_static(u8* blob, const BlobOff& bo, const EncodeArgs& a, u32 T,
                                                  u32 stream_bytes, int lane) {
  const u32 P = (u32)a.P, n = (u32)(a.P * a.G);
  for (u32 i = lane; i < bo.scales - bo.bins; i += 64) blob[bo.bins + i] = i < P ? a.bins.b[i] : (u8)0;
  for (u32 i = bo.scales + 2u * P * T + lane; i < bo.scsum; i += 64) blob[i] = 0;
  for (u32 i = bo.scsum + 4u * P + lane; i < bo.gdir; i += 64) blob[i] = 0;
  for (u32 i = bo.gdir + 8u * n + lane; i < bo.streams; i += 64) blob[i] = 0;
  if (lane < 32) {
    u32 v = 0;
    switch (lane) {
      case 0: v = LMC_BLOB_MAGIC; break;
      case 1: v = LMC_BLOB_VERSION | (LMC_HEADER_BYTES << 16); break;
      case 2: v = (u32)a.dtype; break;
      case 3: v = (u32)a.L; break;
      case 4: v = T; break;
      case 5: v = (u32)a.H; break;
      case 6: v = (u32)a.D; break;
      case 7: v = (u32)a.C; break;
      case 8: v = P; break;
      case 9: v = (u32)a.G; break;
      case 10: v = LMC_LP; break;
      case 11: v = bo.bins; break;
      case 12: v = bo.scales; break;
      case 14: v = bo.gdir; break;  // the stream directory {beg, end}
      case 15: v = bo.streams; break;
      case 16: v = stream_bytes; break;
      case 17: v = bo.streams + stream_bytes; break;
      case 21: v = bo.scsum; break;
      case 22: v = lmc_model_for_dev(T); break;
      default: v = 0;
    }
    reinterpret_cast<u32*>(blob)[lane] = v;
  }
}

#define ENC_WAVES 4           // waves (= group streams) per workgroup (2 and 8 measured 1-2 % slower)
#define ENC_TAB_DWORDS 1056   // 4224 B per wave: histogram [32][64] u16, then (aliased) CDF table [33][64] u16
#define ENC_RING_WORDS 256    // + a staging ring for the renormalisation words (flushed 256 B at a time)
#define ENC_RING_DWORDS ((ENC_RING_WORDS + 64) / 2)  // ... with a 64-word extension: a step never wraps
#define ENC_WAVE_DWORDS (ENC_TAB_DWORDS + ENC_RING_DWORDS)

struct PendingTile {
  int chunk, pg;
  u32 exact, T;
  const u16* out;
};

// In-kernel compaction: where does a stream go?  Single-pass prefix sum over the ALLOCATIONS of the chunk's P*G
// streams (decoupled look-back; lmc_format.h v6: a counts-model stream's allocation is a bound known before it is
// coded, a CDF16 stream's its exact length): allocations are published as soon as they are known, a wave looks back
// over its predecessors' granules until it meets an inclusive prefix and publishes its own inclusive prefix (the
// cumsum + gather of collect_bytes, cachegen_encoder.py:230-238).  Predecessors have lower stream ids: they were taken by
// workgroups dispatched no later than ours, in an earlier or the same round, and never wait on us.
// Exclusive prefix of granule `idx` over the granule array `agg` (decoupled look-back, one wave): walks back 64
// granules at a time until it meets an inclusive prefix; waits while a granule it needs is unpublished.
__device__ __forceinline__ u32 lookback_exclusive(unsigned long long* agg, int idx0, int lane, u32* status) {
  u32 excl = 0;
  if (idx0 > 0) {
    int base = idx0 - 1;
    u32 spins = 0;
    for (;;) {
      const int idx = base - lane;
      const unsigned long long v = idx >= 0 ? agg_load(agg + idx) : ((AGG_P << 62) | 0ull);  // virtual granule -1
      const u32 flag = (u32)(v >> 62);
      const u64 mP = __ballot(flag == (u32)AGG_P), mX = __ballot(flag == (u32)AGG_X);
      const int first = mP ? __builtin_ctzll(mP) : 64;  // nearest predecessor with a full prefix
      const u64 below = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
      if (mX & below) {  // someone we need has not published yet
        if (++spins > (1u << 24)) {
          if (lane == 0) atomicOr(status, LMC_ST_LOOKBACK_TIMEOUT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      excl += wave_sum_u32(lane <= first ? (u32)v : 0u);
      if (mP) break;
      base -= 64;
    }
  }
  return excl;
}

// (The general coder launch only: the fused kernel and the counts-only launch code straight into the blob.)
// A finished stream from its scratch slot to its place in the blob, n16 16-byte pieces by one wave.  `dst[i] = src[i]`
// compiles to "load, wait, store" per KiB (the pointers may alias): 9 dependent memory round trips for a typical
// 8.6 KiB stream with the wave's slot held throughout.  Holding a batch in VGPRs instead costs the whole kernel its
// register budget (8 pieces: 144 bytes of spills in the coder loops, measured +20 % on k_cdf_encode).  So the batch
// lands in LDS: global_load_lds_dwordx4 (gfx950) writes lane l's 16 bytes to M0 + 16 l without a destination
// register, four of them fill the wave's idle 4 KiB table slice, and the pieces go out from there -- 3 round trips.
#ifndef LMC_PLACE_BATCH
#define LMC_PLACE_BATCH 4  // KiB of LDS = loads in flight
#endif
__device__ __forceinline__ void copy_stream16(uint4* dst, const uint4* src, u32 n16_v, u32* lds4k, int lane) {
  typedef __attribute__((address_space(1))) const void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  const u32 n16 = (u32)__builtin_amdgcn_readfirstlane((int)n16_v);  // wave-uniform: scalar loop control
#pragma unroll 1
  for (u32 base = 0; base < n16; base += 64u * LMC_PLACE_BATCH) {
#pragma unroll
    for (int k = 0; k < LMC_PLACE_BATCH; k++) {
      const u32 i = base + 64u * k + lane;
      __builtin_amdgcn_global_load_lds((gptr)(src + (i < n16 ? i : n16 - 1u)), (lptr)(lds4k + 256 * k), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < LMC_PLACE_BATCH; k++) {
      const u32 i = base + 64u * k + lane;
      const uint4 v = *reinterpret_cast<const uint4*>(lds4k + 256 * k + 4 * lane);
      // the placed streams are not read again by this kernel: non-temporal stores
      if (i < n16) __builtin_nontemporal_store(*reinterpret_cast<const u32x4_t*>(&v), reinterpret_cast<LMC_GLOBAL u32x4_t*>((LMC_GLOBAL uint4*)(dst + i)));
      asm volatile("" ::: "memory");  // one piece in registers at a time
    }
  }
  wave_lds_fence();  // the slice is free for its next user
}

// 16-byte zero fill of [p, p + n), n a multiple of 16, by one wave (the slack between a stream and its allocation)
__device__ __forceinline__ void zero_fill16(u8* p, u32 n, int lane) {
  for (u32 i = 16u * (u32)lane; i < n; i += 1024u) *reinterpret_cast<uint4*>(p + i) = make_uint4(0, 0, 0, 0);
}

// A finished stream from its scratch slot to byte `beg` of the chunk's streams section, the zeros up to the end of its
// allocation, and its directory entry {beg, end} (lmc_format.h, v6).  STATIC: the chunk's last stream also writes
// header, static sections and the size word (chunk_total = bytes of the whole streams section).
template <bool STATIC = true>
__device__ __forceinline__ void place_stream(const EncodeArgs& a, const PendingTile& t, u32 beg, u32 alloc, u32 chunk_total,
                                             u32* lds4k, int lane) {
  const int n = a.P * a.G;
  const u32 padded = (t.exact + 15u) & ~15u;
  const BlobOff bo = lmc_blob_off((u32)a.P, t.T, (u32)a.G);
  u8* blob = a.blobs + (long long)t.chunk * a.blob_stride;
  if (lane == 0) {
    u32* d = reinterpret_cast<u32*>(blob + bo.gdir) + 2 * t.pg;
    d[0] = beg;
    d[1] = beg + t.exact;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stream stores have landed before it re-reads them
  copy_stream16(reinterpret_cast<uint4*>(blob + bo.streams + beg), reinterpret_cast<const uint4*>(t.out), padded >> 4, lds4k, lane);
  if (alloc > padded) zero_fill16(blob + bo.streams + beg + padded, alloc - padded, lane);
  if (STATIC && t.pg == n - 1) {  // the last stream knows the chunk's size: header, static sections, size word
    write_blob_static(blob, bo, a, t.T, chunk_total, lane);
    if (lane == 0) a.sizes[t.chunk] = bo.streams + chunk_total;
  }
}

// Symbol of token i (0..31) of a 32-token block whose workspace dwords are in w (k_quantize.h formats):
// bytes: dword i / 4, byte i % 4;  nibbles: dword i / 8, byte (i % 8) % 4, high nibble for i % 8 >= 4.
// v_bfe_u32 as an opaque instruction: given the builtin, the compiler merges the extraction with the table-row
// scaling that follows into shift + and + add (3 VALU); bfe + v_lshl_add_u32 is 2.
template <int POS, int WIDTH>
__device__ __forceinline__ u32 bfe_op(u32 w) {
  u32 r;
  asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "n"(POS), "n"(WIDTH));
  return r;
}
template <bool NIB, int I>
__device__ __forceinline__ u32 sym_of_block(const u32* w) {
  if (NIB) return bfe_op<8 * (I & 3) + 4 * ((I >> 2) & 1), 4>(w[I >> 3]);
  return bfe_op<8 * (I & 3), 8>(w[I >> 2]);
}
// ... and of an arbitrary token t of the plane-chunk column symq (stride C dwords), for the ragged ends
template <bool NIB>
__device__ __forceinline__ u32 sym_of_token(const u32* symq, int C, int t, bool active) {
  if (NIB) {
    const u32 wq = active ? symq[(long long)(t >> 3) * C] : 0u;
    return (wq >> (8 * (t & 3) + 4 * ((t >> 2) & 1))) & 0xfu;
  }
  const u32 wq = active ? symq[(long long)(t >> 2) * C] : 0u;
  return (wq >> (8 * (t & 3))) & 0xffu;
}

// One group stream (chunk, plane, 64-channel group) = global stream id `gid`, by one wave: histogram, counts
// section, CDF table, interleaved rANS into the stream's scratch slot.  `hist` is the wave's table slice of LDS
// (ENC_TAB_DWORDS), `ring` its staging ring (ENC_RING_DWORDS).  Returns the finished stream in `t` (ENCODE).
// This is the CDF16 form (chunk lengths other than 256; lmc_format.h); 256-token chunks take
// encode_group_stream_counts (k_encode_counts.h).
template <bool QUADSYM, bool ENCODE>
__device__ __forceinline__ void encode_group_stream(const EncodeArgs& a, long long gid, u32* hist, u16* const ring,
                                                    int lane, PendingTile& t) {
  u16* tab = reinterpret_cast<u16*>(hist);          // [33][64] u16 CDF, written after hist is in registers
  const int g = (int)(gid % a.G);
  const long long pc = gid / a.G;
  const int p = (int)(pc % a.P);
  const int chunk = (int)(pc / a.P);
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  const u32* symq = QUADSYM ? a.sym4 + ((long long)chunk * a.P + p) * a.sym_stride + c : nullptr;
  const int8_t* symb = QUADSYM ? nullptr : a.sym8 + (long long)p * Tc * a.C + c;

  // ---- pass 1: histogram --------------------------------------------------
  // Blocks of 8 quads (32 tokens).  Every block but the last is full, so the hot path carries no
  // per-token guards; the next block's symbols are loaded while the current one is processed.
#pragma unroll
  for (int i = 0; i < 16; i++) hist[i * 64 + lane] = 0;
  // counter of (symbol s, lane) = u16 at [s][lane], the layout of the CDF table that replaces it: the
  // pair of lanes sharing a dword add 1 and 1 << 16 (a constant per lane, so a symbol costs one bfe and
  // one shift-add); equal symbols in a pair are serialised by the LDS atomic unit.
  u32* const hrow = hist + (lane >> 1);
  const u32 one = 1u << ((lane & 1) * 16);
  // Workspace symbol format (k_quantize.h): byte planes hold 4 tokens per dword, nibble planes (bins <= 17) 8;
  // sym_of_block<NIB, i>(w) is the symbol of token i (0..31) of a 32-token block held in w[0 .. DPB).
  const bool nib = QUADSYM && lmc_sym_nibbles((int)a.bins.b[p]);  // wave-uniform
  auto pass1 = [&](auto nib_tag) {
    constexpr bool NIB = decltype(nib_tag)::value;
    constexpr int DPB = NIB ? 4 : 8;  // dwords per 32-token block
    const int nfull = Tc >> 5;        // full 32-token blocks
    u32 w[DPB], wn[DPB];
    if (nfull > 0) {
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = active ? symq[(long long)j * a.C] : 0u;
    }
    for (int b = 0; b < nfull; b++) {
      if (b + 1 < nfull) {
#pragma unroll
        for (int j = 0; j < DPB; j++) wn[j] = active ? symq[(long long)((b + 1) * DPB + j) * a.C] : 0u;
      }
#pragma unroll
      for (int j = 0; j < DPB; j++) {
        static_for<32 / DPB>([&](auto ktag) {
          constexpr int k = decltype(ktag)::value;
          const u32 sk = bfe_op<(NIB ? 4 : 8) * k, NIB ? 4 : 8>(w[j]);  // any order will do here
          atomicAdd(&hrow[sk * 32], one);  // ds_add_u32 of this lane's half of the dword, bank = lane / 2
        });
      }
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = wn[j];
    }
    for (int t = nfull * 32; t < Tc; t++) {  // ragged tail (< 32 tokens)
      const u32 s = sym_of_token<NIB>(symq, a.C, t, active);
      atomicAdd(&hrow[s * 32], one);
    }
  };
  if (QUADSYM) {
    if (nib) pass1(BoolTag<true>{});
    else pass1(BoolTag<false>{});
  } else {
    for (int t = 0; t < Tc; t++) {
      const u32 s = active ? min((u32)(u8)symb[(long long)t * a.C], 31u) : 0u;
      atomicAdd(&hrow[s * 32], one);
    }
  }
  if (ENCODE) {
    wave_lds_fence();  // every lane's ds_add has landed
    if (g == 0) {  // one wave per (chunk, plane): checksum of the plane's scales (written by k_quantize)
      const BlobOff bo = lmc_blob_off((u32)a.P, (u32)Tc, (u32)a.G);
      u8* const blob0 = a.blobs + (long long)chunk * a.blob_stride;
      const u32 cs = scale_checksum(reinterpret_cast<const u16*>(blob0 + bo.scales) + (long long)p * Tc, (u32)Tc, lane);
      if (lane == 0) reinterpret_cast<u32*>(blob0 + bo.scsum)[p] = cs;
    }
  }
  u32 hreg[16];  // this lane's 32 counts, two per register
#pragma unroll
  for (int i = 0; i < 16; i++) hreg[i] = (u32)tab[(2 * i) * 64 + lane] | ((u32)tab[(2 * i + 1) * 64 + lane] << 16);
  wave_lds_fence();  // hist is dead from here on: tab aliases it

  // ---- the stream opens with its head: the counts, bit-sliced (a chunk of 256 tokens never comes here, so no count
  // saturates: T < 256 keeps them below 256, T > 256 stores them as they are) -------------------------------------
  u16* out = ENCODE ? reinterpret_cast<u16*>(a.scratch + gid * (long long)a.cap) : nullptr;
  u32 head = 0;
  if (ENCODE) {
    u32 pk[16], wor[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pk[i] = active ? hreg[i] : 0u;
    head_or_counts<16>(pk, wor);
    head = head_write<16, 16>(reinterpret_cast<u8*>(out), pk, wor, (u32)a.bins.b[p] - 1u, reinterpret_cast<u32*>(tab), lane);
    wave_lds_fence();  // the words parked in `tab` are dead
    out += head >> 1;
  }

  // ---- CDF ------------------------------------------------------------------
  const u32 T = (u32)Tc;
  // !ENCODE (lmc_calculate_cdf) has no bins: every entry is computed
  cdf_column_to_lds(hreg, T, ENCODE ? (u32)a.bins.b[p] - 1u : 33u, tab, lane);
  if (ENCODE && nib) {
    // Planes with <= 16 symbols use table entries 0 .. 16 only: rows 17 .. 32 get the symbols' FREQUENCIES
    // (row 17 + s = cdf[s + 1] - cdf[s]), so that the token loop reads (start, freq) instead of (start, end)
    // and saves the subtraction.  A lane reads back its own column: program order is enough.
    u32 prev = tab[lane];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 nxt = tab[(i + 1) * 64 + lane];
      tab[(17 + i) * 64 + lane] = (u16)(nxt - prev);
      prev = nxt;
    }
  }
  wave_lds_fence();  // columns are read by other lanes below
  if (ENCODE) {
    // nothing to write: the blob carries the counts (written above), the CDF is a function of them
  } else {
    // lmc_calculate_cdf: the reference's full [P][C][33] layout
    const u32 total = (u32)min(64, a.C - g * 64) * LMC_LP;
    u16* dst = a.cdf_out + ((long long)p * a.C + g * 64) * LMC_LP;
    for (u32 e = lane; e < total; e += 64) {
      const u32 cl = div33(e), s = e - cl * LMC_LP;
      dst[e] = tab[s * 64 + cl];
    }
  }
  if (!ENCODE) return;

  // ---- pass 2: interleaved rANS ---------------------------------------------
  // Idle lanes (channel >= C) see symbol 0 only (w = 0, start 0): starting them at x = 0 keeps them at 0,
  // so they never satisfy the renormalisation test and the hot loop needs no lane predicate.
  u32 x = active ? LMC_RANS_L : 0u;
  u32 wcur = 0;  // wave-uniform word cursor

  // one token: renormalise (append this step's words in ascending lane order), then encode
  // The words of a step (a few dozen bytes) go to a wave-private LDS ring; whenever 128 words have gathered
  // they leave with one coalesced 256-byte store.
  typedef __attribute__((address_space(3))) u16* lds_u16w;
  const u32 ring_addr = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(lds_u16w)ring);
  const u64 full_exec = __builtin_amdgcn_read_exec();
  LMC_GLOBAL u32* const out32 = (LMC_GLOBAL u32*)out;
  u32 flushed = 0;  // words already in global memory (a multiple of 128), wave-uniform
  // The ring has 256 slots and a 64-slot extension: a step's words go to consecutive slots from wcur % 256 on
  // (no wrap inside a step); what ran past slot 255 is brought back to slots 0.. when the upper half is flushed
  // (a step that crosses a multiple of 256 always completes the upper half: at most 127 words are unflushed).
  auto code_token = [&](u32 st, u32 f) {
    // emit <=> x >= f << 16.  Under exec = emitting lanes: rank, slot, store of the low half, x >>= 16.
    const u32 wbase = ring_addr + ((wcur & (ENC_RING_WORDS - 1)) << 1);  // scalar
    u32 t, cnt;
    asm volatile("v_cmpx_ge_u32_sdwa vcc, %[x], %[f] src0_sel:WORD_1 src1_sel:DWORD\n\t"
                 "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                 "s_nop 0\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[wb]\n\t"
                 "ds_write_b16 %[t], %[x]\n\t"
                 "v_lshrrev_b32_e32 %[x], 16, %[x]\n\t"
                 "s_mov_b64 exec, %[full]"
                 : [x] "+v"(x), [t] "=&v"(t), [cnt] "=&s"(cnt)
                 : [f] "v"(f), [wb] "s"(wbase), [full] "s"(full_exec)
                 : "vcc", "scc", "memory");
    wcur += cnt;
    if (wcur - flushed >= 128u) {
      wave_lds_fence();
      if (flushed & 128u) {  // the upper half leaves: bring the words that ran past slot 255 back to slots 0..
        const u32 over = wcur - flushed - 128u;  // < 64
        if ((u32)lane < over) ring[lane] = ring[ENC_RING_WORDS + lane];
      }
      out32[(flushed >> 1) + lane] = reinterpret_cast<const u32*>(ring)[((flushed & (ENC_RING_WORDS - 1)) >> 1) + lane];
      flushed += 128u;
    }
    x = rans_put(x, f, 0x10000u - f, st);
  };
  auto pass2 = [&](auto nib_tag) {
    constexpr bool NIB = decltype(nib_tag)::value;
    constexpr int DPB = NIB ? 4 : 8;
    // ragged head of the descending walk: tokens Tc-1 .. 32*nfull (< 32 of them), one at a time
    const int nfull = Tc >> 5;
    // second table value of a symbol: its frequency (nibble planes: row 17 + s) or the next CDF entry
    constexpr int ROW2 = NIB ? 17 * 64 : 64;
    auto freq_of = [&](u32 lo, u32 v2) -> u32 { return NIB ? v2 : ((v2 - lo) & 0xffffu); };
    // the table's second value, zero-extended AT the ds_read_u16 (the empty asm keeps the compiler from sinking the
    // extension to the use in the next iteration, where it costs a v_and per token)
    auto second = [&](u32 srow) -> u32 {
      u32 v2 = tab[srow * 64 + ROW2 + lane];
      if (NIB) asm("" : "+v"(v2));
      return v2;
    };
    for (int t = Tc - 1; t >= nfull * 32; t--) {
      const u32 s = sym_of_token<NIB>(symq, a.C, t, active);
      const u32 lo = tab[s * 64 + lane], v2 = second(s);
      code_token(lo, freq_of(lo, v2));
    }
    // full 32-token blocks, descending; software pipelined twice over:
    //   - the next block's symbol dwords are loaded while this block is coded,
    //   - the table entries of token t-1 are fetched from LDS before token t is coded.
    if (nfull > 0) {
      u32 w[DPB], wn[DPB];
#pragma unroll
      for (int j = 0; j < DPB; j++) w[j] = active ? symq[(long long)((nfull - 1) * DPB + j) * a.C] : 0u;
      u32 sn = sym_of_block<NIB, 31>(w);
      u32 lo_n = tab[sn * 64 + lane], hi_n = second(sn);
      for (int b = nfull - 1; b >= 0; b--) {
        if (b > 0) {
#pragma unroll
          for (int j = 0; j < DPB; j++) wn[j] = active ? symq[(long long)((b - 1) * DPB + j) * a.C] : 0u;
        }
        static_for<32>([&](auto itag) {
          constexpr int i = 31 - decltype(itag)::value;  // token of the block, descending
          const u32 st = lo_n, f = freq_of(lo_n, hi_n);
          if constexpr (i > 0) {  // entries of the next token of this block
            sn = sym_of_block<NIB, i - 1>(w);
            lo_n = tab[sn * 64 + lane];
            hi_n = second(sn);
          } else {
            if (b > 0) {          // ... or of the first token of the next block
              sn = sym_of_block<NIB, 31>(wn);
              lo_n = tab[sn * 64 + lane];
              hi_n = second(sn);
            }
          }
          code_token(st, f);
        });
#pragma unroll
        for (int j = 0; j < DPB; j++) w[j] = wn[j];
      }
    }
  };
  if (nib) pass2(BoolTag<true>{});
  else pass2(BoolTag<false>{});
  x = active ? x : LMC_RANS_L;  // idle lanes (channel >= C) carry the initial state
  // the words still in the ring (< 128 + 64)
  wave_lds_fence();
  for (u32 k = flushed + lane; k < wcur; k += 64) out[k] = ring[k & (ENC_RING_WORDS - 1)];
  // tail: states, pad, length
  out[wcur + 2 * lane] = (u16)x;
  out[wcur + 2 * lane + 1] = (u16)(x >> 16);
  wcur += 128;
  const u32 exact = head + wcur * 2;  // head is a multiple of 16
  const u32 padw = ((16u - (exact & 15u)) & 15u) >> 1;
  if ((u32)lane < padw) out[wcur + lane] = 0;
  if (lane == 0 && exact + 16 > a.cap) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);
  t.chunk = chunk; t.pg = p * a.G + g; t.exact = exact; t.T = T; t.out = out - (head >> 1);
}

