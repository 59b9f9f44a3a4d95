// k_encode.h -- per-(plane, 64-channel group) histogram -> CDF -> interleaved
// rANS encode, then stream compaction.
//
// Replaces torchac_cuda.calculate_cdf (call site cachegen_encoder.py:287-289,
// in-tree spec :95-126, :175-222), torchac_cuda.encode_fast_new (:255-260) and
// collect_bytes (:225-238).  Stream format: include/lmc_format.h.
//
// One wave = one group stream: lane = channel.  Everything a wave needs is
// wave-private (its LDS slice, its scratch slot), so there is no barrier in
// this kernel; 4 waves share a workgroup only to share the CU.
//
//   pass 1  per-lane histogram in LDS [bin pair][lane] (2 x u16 counters per
//           dword, ds_add, bank = lane: conflict free)
//   CDF     cdf[i] = RNE(n_i * 65504 / T) + i, exact integer arithmetic
//           -> staged in LDS and written to the blob's cdf section as whole
//              16-byte vectors;  coder table tab[s][lane] = freq<<16 | start
//   pass 2  tokens T-1..0: renormalise (ballot + mbcnt append of 16-bit words,
//           ascending lane order), then x = (x/f << 16) + x%f + start
//   tail    64 states, zero pad to 16 B, exact length -> glen
#pragma once
#include "lmc_device.h"

struct EncodeArgs {
  // symbols
  const u32* sym4;      // QUADSYM: [nchunks][P][TQ][C]
  const int8_t* sym8;   // !QUADSYM: [P][T][C]
  int tok_begin, tok_end, chunk_tokens, nchunks;
  int P, C, G, TQ;
  // outputs
  u8* blobs;            // ENCODE: blob i at blobs + i*blob_stride (cdf section written here)
  long long blob_stride;
  u16* cdf_out;         // !ENCODE: [P][C][33]
  u8* scratch;          // [nchunks*P*G][cap] padded group streams
  u32 cap;              // bytes per scratch slot
  u32* glen;            // [nchunks*P*G] exact stream bytes
  u32* status;
};

#define ENC_WAVE_DWORDS 3104  // tab 32*64 dwords (8 KiB, aliases hist 16*64) + cdf stage 1056 dwords (4224 B)

template <bool QUADSYM, bool ENCODE>
__global__ __launch_bounds__(256) void k_cdf_encode(EncodeArgs a) {
  __shared__ __attribute__((aligned(16))) u32 lds_all[4 * ENC_WAVE_DWORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gid = (long long)blockIdx.x * 4 + wave;
  const long long ngroups_total = (long long)a.nchunks * a.P * a.G;
  if (gid >= ngroups_total) return;
  u32* tab = lds_all + wave * ENC_WAVE_DWORDS;  // [32][64]
  u32* hist = tab;                              // [16][64] (dead before tab is written)
  u16* stage = reinterpret_cast<u16*>(tab + 32 * 64);  // [64][33] u16

  const int g = (int)(gid % a.G);
  const long long pc = gid / a.G;
  const int p = (int)(pc % a.P);
  const int chunk = (int)(pc / a.P);
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);
  const int TQc = (Tc + 3) >> 2;
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  const u32* symq = QUADSYM ? a.sym4 + ((long long)chunk * a.P + p) * a.TQ * a.C + c : nullptr;
  const int8_t* symb = QUADSYM ? nullptr : a.sym8 + (long long)p * Tc * a.C + c;

  // ---- pass 1: histogram --------------------------------------------------
#pragma unroll
  for (int i = 0; i < 16; i++) hist[i * 64 + lane] = 0;
  if (QUADSYM) {
    for (int qb = 0; qb < TQc; qb += 8) {
      u32 w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = (active && qb + j < TQc) ? symq[(long long)(qb + j) * a.C] : 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (qb + j >= TQc) break;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (4 * (qb + j) + k >= Tc) break;
          u32 s = (w[j] >> (8 * k)) & 0xffu;
          if (active) atomicAdd(&hist[(s >> 1) * 64 + lane], 1u << ((s & 1u) * 16));
        }
      }
    }
  } else {
    for (int t = 0; t < Tc; t++) {
      if (active) {
        u32 s = min((u32)(u8)symb[(long long)t * a.C], 31u);
        atomicAdd(&hist[(s >> 1) * 64 + lane], 1u << ((s & 1u) * 16));
      }
    }
  }
  u32 hreg[16];
#pragma unroll
  for (int i = 0; i < 16; i++) hreg[i] = hist[i * 64 + lane];
  wave_lds_fence();  // hist is dead from here on: tab aliases it

  // ---- CDF ------------------------------------------------------------------
  const u32 T = (u32)Tc;
  const u32 magic = (T == 1u) ? 0xffffffffu : (u32)(0x100000000ull / T);
  {
    u32 n = 0, prev = 0;
#pragma unroll
    for (int i = 0; i <= 32; i++) {
      u32 ci = (rne_div_u32(n * LMC_CDF_SCALE, T, magic) + (u32)i) & 0xffffu;
      stage[lane * LMC_LP + i] = (u16)ci;
      if (i >= 1) tab[(i - 1) * 64 + lane] = (((ci - prev) & 0xffffu) << 16) | prev;
      prev = ci;
      if (i < 32) n += (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
    }
  }
  wave_lds_fence();  // stage rows are read by other lanes below
  {
    const int nvalid = min(64, a.C - g * 64);
    const int n16 = nvalid * (LMC_LP * 2) / 16;  // C % 8 == 0 -> whole vectors
    u16* dst;
    if (ENCODE) {
      BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G);
      dst = reinterpret_cast<u16*>(a.blobs + (long long)chunk * a.blob_stride + bo.cdf);
    } else {
      dst = a.cdf_out;
    }
    dst += ((long long)p * a.C + g * 64) * LMC_LP;
    const uint4* s4 = reinterpret_cast<const uint4*>(stage);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (int i = lane; i < n16; i += 64) d4[i] = s4[i];
  }
  if (!ENCODE) return;

  // ---- pass 2: interleaved rANS ---------------------------------------------
  u16* out = reinterpret_cast<u16*>(a.scratch + gid * (long long)a.cap);
  u32 x = LMC_RANS_L;
  u32 wcur = 0;  // wave-uniform word cursor
  for (int qb = (TQc - 1) & ~7; qb >= 0; qb -= 8) {
    u32 w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = (active && qb + j < TQc) ? symq[(long long)(qb + j) * a.C] : 0u;
#pragma unroll
    for (int j = 7; j >= 0; j--) {
      if (qb + j >= TQc) continue;
#pragma unroll
      for (int k = 3; k >= 0; k--) {
        if (4 * (qb + j) + k >= Tc) continue;
        const u32 s = (w[j] >> (8 * k)) & 0xffu;
        const u32 e = tab[s * 64 + lane];
        const u32 f = e >> 16, st = e & 0xffffu;
        const bool emit = active && (x >= (f << 16));
        const u64 mask = __ballot(emit);
        if (mask) {
          const u32 rank = lane_rank(mask);
          if (emit) {
            out[wcur + rank] = (u16)x;
            x >>= 16;
          }
          wcur += (u32)__popcll(mask);
        }
        if (active) {
          u32 q, r;
          divmod_est(x, f, q, r);
          x = (q << 16) + r + st;
        }
      }
    }
  }
  // tail: states, pad, length
  out[wcur + 2 * lane] = (u16)x;
  out[wcur + 2 * lane + 1] = (u16)(x >> 16);
  wcur += 128;
  const u32 exact = wcur * 2;
  const u32 padw = ((16u - (exact & 15u)) & 15u) >> 1;
  if ((u32)lane < padw) out[wcur + lane] = 0;
  if (lane == 0) {
    a.glen[gid] = exact;
    if (exact + 16 > a.cap) atomicOr(a.status, LMC_ST_STREAM_OVERFLOW);
  }
}

// ---------------------------------------------------------------------------
// Per chunk: exclusive scan of the 16-B padded group lengths -> gend table,
// stream offsets for the pack kernel, header, bins, zeroed section pads, size.
// Replaces the cumsum/roll of collect_bytes (cachegen_encoder.py:230-236) and
// the cumsum the decoder would otherwise redo (cachegen_decoder.py:62-64).
struct ScanArgs {
  u8* blobs;
  long long blob_stride;
  const u32* glen;   // [nchunks][P*G]
  u32* goff;         // [nchunks][P*G] start offset (relative to streams) of each group
  u32* sizes;        // [nchunks]
  BinsArg bins;
  int tok_begin, tok_end, chunk_tokens;
  int L, H, D, P, C, G, dtype;
};

__global__ __launch_bounds__(1024) void k_scan_finalize(ScanArgs a) {
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  const int chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const u32 T = (u32)min(a.chunk_tokens, a.tok_end - tok0);
  const int n = a.P * a.G;
  u8* blob = a.blobs + (long long)chunk * a.blob_stride;
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G);
  u32* gend = reinterpret_cast<u32*>(blob + bo.gend);
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const u32 len = i < n ? a.glen[(long long)chunk * n + i] : 0u;
    const u32 padded = (len + 15u) & ~15u;
    u32 incl = padded;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      u32 v = (u32)__shfl_up((int)incl, off);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    const u32 carry = carry_s;
    const u32 excl = carry + wbase + incl - padded;
    if (i < n) {
      gend[i] = excl + len;
      a.goff[(long long)chunk * n + i] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + incl;
    __syncthreads();
  }
  const u32 stream_bytes = carry_s;
  // bins + zeroed pads (the oracle memsets the static region)
  for (u32 i = tid; i < bo.scales - bo.bins; i += 1024) blob[bo.bins + i] = i < (u32)a.P ? a.bins.b[i] : (u8)0;
  for (u32 i = bo.scales + 2u * a.P * T + tid; i < bo.cdf; i += 1024) blob[i] = 0;
  for (u32 i = bo.gend + 4u * n + tid; i < bo.streams; i += 1024) blob[i] = 0;
  if (tid < 32) {
    u32 v = 0;
    switch (tid) {
      case 0: v = LMC_BLOB_MAGIC; break;
      case 1: v = LMC_BLOB_VERSION | (LMC_HEADER_BYTES << 16); break;
      case 2: v = (u32)a.dtype; break;
      case 3: v = (u32)a.L; break;
      case 4: v = T; break;
      case 5: v = (u32)a.H; break;
      case 6: v = (u32)a.D; break;
      case 7: v = (u32)a.C; break;
      case 8: v = (u32)a.P; break;
      case 9: v = (u32)a.G; break;
      case 10: v = LMC_LP; break;
      case 11: v = bo.bins; break;
      case 12: v = bo.scales; break;
      case 13: v = bo.cdf; break;
      case 14: v = bo.gend; break;
      case 15: v = bo.streams; break;
      case 16: v = stream_bytes; break;
      case 17: v = bo.streams + stream_bytes; break;
      default: v = 0;
    }
    reinterpret_cast<u32*>(blob)[tid] = v;
  }
  if (tid == 0) a.sizes[chunk] = bo.streams + stream_bytes;
}

// Copy each padded group stream from its scratch slot to its final place.
// Replaces the fancy-index gather of collect_bytes (cachegen_encoder.py:237-238).
struct PackArgs {
  u8* blobs;
  long long blob_stride;
  const u8* scratch;
  u32 cap;
  const u32* glen;
  const u32* goff;
  int tok_begin, tok_end, chunk_tokens;
  int P, C, G;
  long long ngroups_total;
};

__global__ __launch_bounds__(256) void k_pack_streams(PackArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gid = (long long)blockIdx.x * 4 + wave;
  if (gid >= a.ngroups_total) return;
  const int n = a.P * a.G;
  const int chunk = (int)(gid / n);
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const u32 T = (u32)min(a.chunk_tokens, a.tok_end - tok0);
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G);
  const u32 n16 = ((a.glen[gid] + 15u) & ~15u) >> 4;
  const uint4* src = reinterpret_cast<const uint4*>(a.scratch + gid * (long long)a.cap);
  uint4* dst = reinterpret_cast<uint4*>(a.blobs + (long long)chunk * a.blob_stride + bo.streams + a.goff[gid]);
  for (u32 i = lane; i < n16; i += 64) dst[i] = src[i];
}
