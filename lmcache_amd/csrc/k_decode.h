// k_decode.h -- per-(plane, 64-channel group) interleaved rANS decode fused with
// dequantise, 16-bit cast and the scatter into the caller's KV layout
// (SURVEY.md section 8a rows a13-a16, a18, a19).
//
// Replaces decode_chunk + torchac_cuda.decode_fast_prefsum
// (lmcache/storage_backend/serde/cachegen_decoder.py:51-66), the uint8->fp32
// inflation (:95-104), do_dequantize (:24-35), stack/reshape/permute/.to(16-bit)
// (:177-200) and the engine's torch.cat over chunks (cache_engine.py:362-368):
// the decoded value of (plane, token, channel) is written exactly once, in its
// final place.
//
//   t = ((q - C) / C) * max1      three rounded fp32 ops; (q-C)/C comes from a
//                                  32-entry per-plane LUT built with the same
//                                  fp32 subtraction and IEEE division
//   out = RNE16(t)
//
// One wave = one group stream, lane = channel.  Per token: two search levels on three register pivots, then
//   planes with <= 16 symbols (16/17 bins, most planes): ONE ds_read_b128 of the lane's quarter of packed
//     entries (cdf << 16 | freq), the last two levels as exec-predicated moves among them, and the state
//     update straight from the selected entry (v_mad_u32_u16 with op_sel) -- one LDS round trip in the search;
//   other planes: three dependent probes of the lane's u16 column ([entry][lane], bank = lane/2) and the
//     symbol's two entries;
// then the pop of 16-bit words under exec = renormalising lanes (mbcnt rank, one ds_read_u16).  The words come
// from a 256-word LDS ring in stream order, refilled a block (128 words, one dword per lane) ahead of the
// consumer, so no global-memory latency sits on the per-token dependency chain.  The dequantised value leaves
// through a raw buffer store (row = scalar offset, channel = vector offset; idle lanes fall outside the
// descriptor's range).  4.9 KiB of LDS per wave -> 8 waves per SIMD.
#pragma once
#include "k_head.h"

// lmc_tok_off for the decoder's scatter: a block size that is a power of two (every one vLLM offers) costs a shift and a
// mask instead of a 32-bit division per lane and 64 tokens (wave-uniform branch).  Kept apart from lmc_tok_off: the same
// lines there cost k_encode_fused ten more spilled SGPRs.
__device__ __forceinline__ long long dec_tok_off(const KvAddr& a, int t) {
  if (!a.slot_mapping) return (long long)t * a.stride_token;
  const u32 s = (u32)a.slot_mapping[t], bs = (u32)a.block_size;
  u32 b, w;
  if ((bs & (bs - 1u)) == 0u) {
    b = s >> (u32)__builtin_ctz(bs);
    w = s & (bs - 1u);
  } else {
    b = s / bs;
    w = s - b * bs;
  }
  return (long long)b * a.stride_block + (long long)w * a.stride_token;
}

struct DecodeArgs {
  const u8* blobs;
  long long blob_stride;
  const u8* const* blob_ptrs;  // device array [nchunks] of blob addresses, or NULL (then blobs + i * blob_stride)
  int layer_begin, layer_count; // the layers to decode: planes layer_begin .. +layer_count of K and of V
  int nchunks;
  KvAddr dst;          // SYMOUT=false
  int dst_tok0, chunk_tokens;
  int8_t* sym_out;     // SYMOUT=true: [P][T][C] (single blob)
  int P, C, G;
  u32* status;
  // pack (lmc_format.h): `blobs` / `blob_stride` are the static slots, the streams of plane (layer, kv) of chunk c lie at
  // seg_streams + seg_off[(kv L + layer) seg_n + c]
  const unsigned long long* seg_off;
  const u8* seg_streams;
  int seg_n;
};

// LDS of a workgroup: the four waves' dequantisation LUTs (32 x f32 = 128 B each) FIRST -- their byte addresses stay
// below 2^10 and fit into the packed table entries of the counts model -- then per wave: cdfT [33][64] u16 (4224) |
// word ring: 256 x u16 + 64-word mirror (640)
#ifndef LMC_DEC_NT_IN
#define LMC_DEC_NT_IN 1  // the stream words are read once: non-temporal loads (decode 0.963 -> 0.950 ms, same box)
#endif
#ifndef DEC_WAVES
#define DEC_WAVES 1  // waves (= group streams) per workgroup: ONE (round 6) -- a slot is free again the moment its stream is done, not when the slowest of four is (resident waves 89 -> 97 %, decode -3 %: profiles/r06_timelines.md)
#endif
#ifndef LMC_DEC_RING_AGPR
#define LMC_DEC_RING_AGPR 1  // the word ring's in-flight block waits in AGPR a0 behind s_waitcnt vmcnt(N): see block_request
#endif
// a0 is NOT declared to the compiler: a kernel that uses AGPRs gets its 64 registers split 32 VGPRs / 32 AGPRs by this
// LLVM, and the decoder needs 53.  Instead every asm that touches a0 clobbers v59: the kernel then has exactly 60 VGPRs,
// its allocation is 64 registers (granule 8) and the accumulation offset 60 -- a0 is physical register 60, inside the
// wave's allocation and never touched by compiled code.  native.build() refuses a library whose k_decode kernels do not
// report 60 VGPRs and 0 AGPRs (with 61..64 VGPRs a0 would lie OUTSIDE the allocation).
#define LMC_DEC_A0_CLOBBER "v59"
#define DEC_CDF_BYTES 4224
#define DEC_RING_WORDS 256
#define DEC_RING_BYTES (2 * (DEC_RING_WORDS + 64))
#define DEC_LUT_BYTES 128
#define DEC_WAVE_BYTES (DEC_CDF_BYTES + DEC_RING_BYTES)


__device__ __forceinline__ u64 uniform_ptr(const void* p) {  // a pointer every lane holds -> SGPR pair
  const u64 v = (u64)p;
  const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
  return ((u64)hi << 32) | (u64)lo;
}

#ifdef LMC_EXP_TIMELINE  // experiments only: time stamps of every stream's wave (tools/probes/decode_timeline.hip / .py)
__device__ unsigned long long g_decode_timeline[65536 * 4];
#define LMC_DTL(k, v) do { if (lane == 0 && gid < 65536u) g_decode_timeline[gid * 4u + (k)] = (v); } while (0)
#else
#define LMC_DTL(k, v) do { } while (0)
#endif
template <bool SYMOUT, int DT_OUT, bool PAGED>
__global__ __launch_bounds__(64 * DEC_WAVES) __attribute__((amdgpu_waves_per_eu(8))) void k_decode(DecodeArgs a) {
  __shared__ __attribute__((aligned(16))) u8 lds_all[DEC_WAVES * (DEC_LUT_BYTES + DEC_WAVE_BYTES)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform -> SGPRs
  // (32-bit work-item arithmetic: lmc_api.hip rejects a launch of 2^31 streams or more, and a 64-bit division costs
  // more than a hundred instructions per stream)
  u32 gid = blockIdx.x * (u32)DEC_WAVES + (u32)wave;
  const int n = 2 * a.layer_count * a.G;  // streams of a chunk in this launch
  if (gid >= (u32)a.nchunks * (u32)n) return;
#ifndef LMC_DEC_XCD_MAP
#define LMC_DEC_XCD_MAP 1
#endif
  if constexpr (LMC_DEC_XCD_MAP && DEC_WAVES == 1 && !PAGED && !SYMOUT) {
    // (round 6, late) Workgroups go to the eight XCDs round-robin, so the G streams of a plane -- consecutive work items,
    // which write the G 128-byte pieces of the same rows, read the same scales and header -- landed on eight different L2s.
    // Within every tile of 8 G consecutive work items, XCD x (workgroups x, x + 8, ...) takes plane x of the tile: a
    // plane's streams share one L2.  The launch's last, partial tile keeps the plain order.  Only where a token's row is
    // one run of bytes (heads back to back: contiguous decode -1.5 %); with the pieces of a row kilobytes apart -- paged
    // NHBD blocks -- the same map costs 3 - 5 % (profiles/r06_experiments.md section 13), so those keep the plain order.
    const u32 tile = 8u * (u32)a.G, t0 = gid / tile * tile;
    if (a.dst.stride_head == (long long)a.dst.D && t0 + tile <= (u32)a.nchunks * (u32)n) {
      const u32 i = gid - t0;
      gid = t0 + (i & 7u) * (u32)a.G + (i >> 3);
    }
  }
  LMC_DTL(0, (unsigned long long)wall_clock64());
  u8* wl = lds_all + DEC_WAVES * DEC_LUT_BYTES + wave * DEC_WAVE_BYTES;
  u16* cdfT = reinterpret_cast<u16*>(wl);                           // [33][64]: entry-major, bank = lane/2
  u16* ring = reinterpret_cast<u16*>(wl + DEC_CDF_BYTES);           // stream words, word j at slot j % 256; slots 256..319 mirror 0..63
  float* lut = reinterpret_cast<float*>(lds_all + wave * DEC_LUT_BYTES);  // (q - C) / C

  const int chunk = (int)(gid / (u32)n);
  const int r = (int)(gid - (u32)chunk * (u32)n);
  const int pidx = r / a.G, g = r - pidx * a.G;
  // K planes of the layer range first, then their V planes (plane p = kv * L + layer)
  const int p = pidx < a.layer_count ? a.layer_begin + pidx : (a.P >> 1) + a.layer_begin + pidx - a.layer_count;
  const int pg = p * a.G + g;
  const u8* blob = a.blob_ptrs ? (const u8*)uniform_ptr(a.blob_ptrs[chunk]) : a.blobs + (long long)chunk * a.blob_stride;
  // The header's 32 words with ONE coalesced load (lane i holds word i), handed out by v_readlane: all control flow
  // below is scalar, and the validity test -- a chain of || over nine header words -- no longer walks through nine
  // dependent round trips to memory (round 6: the prologue was 15 us of a wave's 95, most of it this chain).
  const u32 hv = ((const LMC_GLOBAL u32*)blob)[lane & 31];
  // Issued with it, before anything waits: the plane's bin count (at a fixed offset) and -- blobs in an arena or a
  // pack's static slots, where a full-length chunk's directory offset lies inside the slot whatever this chunk's
  // length turns out to be -- the directory entry at the offset a chunk of chunk_tokens tokens has it at.  A shorter
  // chunk (a ragged last one) reads its entry again below.  (Blobs behind a pointer table end where they end: no
  // speculative read there.)
  const u32 bins_p = (u32)((const LMC_GLOBAL u8*)blob)[LMC_HEADER_BYTES + p];
  const BlobOff bo_spec = lmc_blob_off((u32)a.P, (u32)a.chunk_tokens, (u32)a.G);
  const bool spec = !SYMOUT && !a.blob_ptrs && (unsigned long long)bo_spec.streams <= (unsigned long long)a.blob_stride;
  u32x2_t gd_spec = {0u, 0u};
  u32 sbeg_spec = 0;
  if (spec) {
    gd_spec = *(const LMC_GLOBAL u32x2_t*)(blob + bo_spec.gdir + 8u * (u32)pg);
    if (a.seg_off) sbeg_spec = *(const LMC_GLOBAL u32*)(blob + bo_spec.gdir + 8u * (u32)(p * a.G));
  }
  auto hdw = [&](int i) { return (u32)__builtin_amdgcn_readlane((int)hv, i); };
  const u32 T = hdw(4);
  const u32 src_dtype = hdw(2);
  const bool counts_model = hdw(22) == LMC_MODEL_COUNTS;  // wave-uniform: the coder ran on the symbol counts (2 <= T <= 256)
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.G);
  if (hdw(0) != LMC_BLOB_MAGIC || (hdw(1) & 0xffffu) != LMC_BLOB_VERSION || hdw(7) != (u32)a.C || T == 0u || T > 65535u ||
      !lmc_model_valid_dev(hdw(22), T) ||  // (the header's model is trusted: a round-3/4 blob of a short chunk is CDF16)
      hdw(8) != (u32)a.P || hdw(15) != bo.streams ||
      // every section offset is a function of the fields checked above; the blob must also fit its slot
      (!SYMOUT && (T > (u32)a.chunk_tokens ||
                   (unsigned long long)hdw(a.seg_off ? 15 : 17) > (unsigned long long)a.blob_stride))) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_HEADER);
    return;
  }
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  // ---- where the stream lies (directory entry {beg, end}, lmc_format.h v6) ------------------------------------------
  const u32* gdir = reinterpret_cast<const u32*>(blob + bo.gdir);
  const bool spec_ok = spec && bo.gdir == bo_spec.gdir;  // wave-uniform
  u32x2_t gd = gd_spec;
  if (!spec_ok) gd = *(const LMC_GLOBAL u32x2_t*)(const LMC_GLOBAL u32*)(gdir + 2 * pg);
  const u32 start = (u32)__builtin_amdgcn_readfirstlane((int)gd.x);
  const u32 end = (u32)__builtin_amdgcn_readfirstlane((int)gd.y);
  const u8* sbytes = blob + bo.streams + start;
  bool seg_bad = false;
  if (a.seg_off) {  // pack: this plane's streams are a segment of their own, wave-uniform addresses
    const long long si = (long long)p * a.seg_n + chunk;  // pack v3: segments in plane order (K planes, then V planes)
    const unsigned long long so = uniform_ptr((const void*)a.seg_off[si]), se = uniform_ptr((const void*)a.seg_off[si + 1]);
    const u32 sbeg = (u32)__builtin_amdgcn_readfirstlane((int)(spec_ok ? sbeg_spec : gdir[2 * (p * a.G)]));  // the plane's first stream
    seg_bad = start < sbeg || se < so || (unsigned long long)end - sbeg > se - so;
    sbytes = a.seg_streams + so + (start - sbeg);
  }
  // 64-bit comparisons: a corrupt directory entry must not wrap its way past the bounds
  if (seg_bad || (start & 15u) || (unsigned long long)end < (unsigned long long)start + 256ull ||
      (unsigned long long)bo.streams + (unsigned long long)end > (unsigned long long)hdw(17)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
    return;
  }

  // ---- the stream's head: symbol counts of this group -> the CDF table in LDS, [entry][lane] u16 ---------------------
  // The head stores the counts of every channel's symbols 0 .. nsym-1 (nsym = bins - 1) bit-sliced (k_head.h); a lane
  // takes its own channel's counts and turns them into its column of the table with the encoder's integer arithmetic.
  const u32 nsym = min(31u, max(3u, (u32)__builtin_amdgcn_readfirstlane((int)bins_p) - 1u));
  // the lanes' final states (the stream's last 256 bytes: their place does not depend on the head's size): requested
  // here, together with the head
  const LMC_GLOBAL u16* const xw = (const LMC_GLOBAL u16*)sbytes + ((end - start) >> 1) - 128u + 2u * (u32)lane;
  const u32 x_lo = xw[0], x_hi = xw[1];
  u32 head_bytes;
  {
    u32 cv[32];
    head_bytes = head_read<32>(sbytes, end - start - 256u, nsym, reinterpret_cast<u32*>(cdfT), cv, lane);
    if (head_bytes == 0u) {
      if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
      return;
    }
    const bool bytes1 = T <= 256u;  // wave-uniform: a count of 256 was stored as 255
    u32 hreg[16];  // this lane's 32 counts, two per register
#pragma unroll
    for (int i = 0; i < 16; i++) hreg[i] = cv[2 * i] | (cv[2 * i + 1] << 16);
    if (bytes1) {  // the counts of a channel sum to T
      u32 both = 0;  // the two halves add up side by side: no count is above 255 here, so neither sum passes 16 bits
#pragma unroll
      for (int i = 0; i < 16; i++) both += hreg[i];
      const u32 sum = (both & 0xffffu) + (both >> 16);
      const u32 deficit = active ? T - sum : 0u;  // 0 or 1 in a well-formed blob
      if (__ballot(deficit != 0u)) {
        if (counts_model) {
          // the counts model codes such a channel with 255 and a count of 1 on symbol 0 -- on symbol 1 if the 255 is
          // symbol 0's (lmc_counts_model): the missing unit goes there
          const bool first = (hreg[0] & 0xffffu) == 255u;
          hreg[0] += first ? (deficit & 0xffffu) << 16 : (deficit & 0xffffu);
        } else {
#pragma unroll
          for (int i = 0; i < 16; i++) {
            if ((hreg[i] & 0xffffu) == 255u) hreg[i] += deficit & 0xffffu;
            else if ((hreg[i] >> 16) == 255u) hreg[i] += (deficit & 0xffffu) << 16;
          }
        }
      }
      if (counts_model && T != LMC_COUNTS_T) {
        // chunks below 256 tokens (round 5): the MODEL counts are the counts scaled to a sum of 256 along the cumulative
        // sum, n[s] = floor(256 C_s / T) - floor(256 C_(s-1) / T) (lmc_format.h: lmc_counts_model; the quotient by one
        // multiply-high, exact for C <= T <= 256), and a channel with one symbol -- n = 256 -- becomes 255 + 1
        const u32 magic = lmc_counts_scale_magic_dev(T);
        u32 cum = 0, prev = 0, big = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          cum = min(cum + (hreg[i] & 0xffffu), T);  // (min: a damaged head cannot push a count past 256)
          const u32 n0 = __umulhi(cum << 8, magic);
          cum = min(cum + (hreg[i] >> 16), T);
          const u32 n1 = __umulhi(cum << 8, magic);
          hreg[i] = (n0 - prev) | ((n1 - n0) << 16);
          prev = n1;
          big |= hreg[i];
        }
        const bool first = (hreg[0] & 0xffffu) == 256u;
#pragma unroll
        for (int i = 0; i < 16; i++) hreg[i] -= (hreg[i] >> 8) & 0x00010001u;  // 256 -> 255 (no other count has bit 8)
        if (big & 0x01000100u) hreg[0] += first ? 0x10000u : 1u;
      }
    }
    wave_lds_fence();  // the staging is dead (every lane holds its counts): the table takes its place
    if (counts_model) {
      // LMC_MODEL_COUNTS: freq = 2 * count, start = 2 * (symbols below), total 2^9.
      if (nsym <= 16u) {
        // quarters of packed entries: start in bits 23 .. 31 (start is even), and in the 23 bits below
        //     0x7ffffe - (3 - i % 4) << 20 - &lut[symbol] << 10 - freq
        // (the LUTs lie in the first KiB of LDS: 2 + 10 + 10 bits).  A search key is slot << 23 | 0x7ffffe, so
        // key - entry, for an entry that is <= key, is (slot - start) << 23 | (3 - i % 4) << 20 | &lut << 10 | freq
        // without a borrow -- and UNSIGNED, the smallest of the four differences of a quarter belongs to the largest
        // entry <= key (an entry above the key wraps to more than the key): the token loop takes start, freq and
        // the LUT address out of one v_min3 / v_min, no selection among the entries.  (Entries of symbols that do
        // not occur share their successor's start: i % 4 makes the later one the larger.)
        // Symbols behind the last one that occurs would start at 2^9: all ones, above every search key.
        // (A lane without a channel stored zeros: all its entries start at 0, whichever the search finds has freq 0 and
        // a LUT address of this wave, its state stays 0 and never renormalises -- no special column for it.)
        u32* tab32 = reinterpret_cast<u32*>(cdfT);
        const u32 lut0 = (u32)(size_t)(const __attribute__((address_space(3))) float*)lut;
        if (lut0 + DEC_LUT_BYTES > 1024u) __builtin_trap();  // lds_all is the kernel's only LDS object: offset 0
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const u32 ci = (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
          u32 ent = acc >= 256u ? 0xffffffffu : ((acc << 24) | (0x7ffffeu - (((3u - ((u32)i & 3u)) << 20) | ((lut0 + 4u * (u32)i) << 10) | (ci << 1))));
          tab32[(i >> 2) * 256 + lane * 4 + (i & 3)] = ent;
          acc += ci;
        }
      } else {
        // u16 column of starts [entry][lane], entry 32 = 2^9 (a CDF out of 512: the narrow search runs on it unchanged)
        u32 acc = 0;
#pragma unroll
        for (int i = 0; i <= 32; i++) {
          cdfT[i * 64 + lane] = active ? (u16)min(2u * acc, 512u) : (u16)i;
          if (i < 32) acc += (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
        }
      }
    } else if (nsym <= 16u) {
      // planes with at most 16 symbols (16 / 17 bins: most of them): 16 entries of 32 bits in four quarters,
      // tab32[quarter][lane][4], entry i = cdf[i] << 16 | (cdf[i + 1] - cdf[i]): one ds_read_b128 brings a quarter
      cdf_column_to_lds_wide(hreg, T, nsym, reinterpret_cast<u32*>(cdfT), lane);
      if (!active) {  // idle lanes: any strictly increasing column keeps the search in range
#pragma unroll
        for (int i = 0; i < 16; i++) reinterpret_cast<u32*>(cdfT)[(i >> 2) * 256 + lane * 4 + (i & 3)] = ((u32)i << 16) | 1u;
      }
    } else {
      cdf_column_to_lds(hreg, T, nsym, cdfT, lane);
      if (!active) {
#pragma unroll
        for (int i = 0; i < 33; i++) cdfT[i * 64 + lane] = (u16)i;
      }
    }
  }
  // ---- dequantisation LUT; the per-token scales are fetched 64 tokens at a time inside the loop ----
  const u16* scl = reinterpret_cast<const u16*>(blob + bo.scales) + (long long)p * T;
  if (!SYMOUT && g == 0) {  // the scales are the one section whose damage the coder cannot see: the plane's first wave
                            // checks their checksum (every launch decodes all the groups of the planes it takes).
                            // The other G - 1 waves do not wait for the verdict: on LMC_ST_BAD_SCALES the destination's
                            // contents are UNDEFINED (include/lmc_hip.h says so; the engine turns any non-zero status
                            // into a miss and the caller recomputes those tokens).  Checking in every wave was measured
                            // again in round 5: +1.5 % on the 16 k decode, alternating builds on one box.
    const u32 want = (u32)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const u32*>(blob + bo.scsum)[p]);
    if (scale_checksum(scl, T, lane) != want) {
      if (lane == 0) atomicOr(a.status, LMC_ST_BAD_SCALES);
      return;
    }
  }
  if (!SYMOUT && lane < 32) {
    const float Cf = (float)((int)bins_p / 2 - 1);
    const float v = (float)lane - Cf;
    lut[lane] = v / Cf;
  }

  LMC_DTL(1, (unsigned long long)wall_clock64());
  // ---- stream ----------------------------------------------------------------
  const u16* words = reinterpret_cast<const u16*>(sbytes + head_bytes);
  const u32 nwords = ((end - start - head_bytes) >> 1) - 128u;  // 16-bit words in front of the 64 states
  u32 x = x_lo | (x_hi << 16);
  // The decoder consumes the stream from its end: after `e` = number of words not yet consumed, a token whose
  // cnt lanes renormalise takes words [e - cnt, e), ascending with the lane (the encoder's append order).  The
  // words are staged in STREAM ORDER in a 256-word LDS ring, word j at slot j % 256, in blocks of 128 words
  // (block b = words [128 b, 128 b + 128), one coalesced dword per lane): two blocks are resident, the loads of
  // next lower one are REQUESTED into a register the moment the block above them has been written to the ring, and
  // WRITTEN to the ring when e <= 128 top (top = upper resident block: it is consumed; a token takes at most 64 words,
  // so the block below covers the next token alone) -- a whole block, nine tokens or so, later (round 6; until then the
  // request went out at e <= 128 top + 64, four or five tokens ahead, and under the kernel's own 2.6 TB/s of stores
  // the words were late: -6 % decode time, and one ring event per block instead of two):
  // their latency is off the per-token dependency chain.  Slots 256..319 mirror slots 0..63, so a token's words
  // are at consecutive slots whatever e.  A corrupt stream cannot leave the ring (slot index masked, rank < 64,
  // loads guarded); the final state / count check reports it.
  typedef __attribute__((address_space(3))) u32* lds_u32w;
  const lds_u32w ring32 = (lds_u32w)reinterpret_cast<u32*>(ring);
  const LMC_GLOBAL u32* const words32 = (const LMC_GLOBAL u32*)words;  // streams start 16-byte aligned
  auto block_load = [&](int b) -> u32 {
    const int d = 64 * b + lane;  // dword = words 2d, 2d + 1; the 128 state words that follow belong to the stream too
#if LMC_DEC_NT_IN
    return (b >= 0 && (u32)(2 * d + 1) < nwords + 128u) ? __builtin_nontemporal_load(words32 + d) : 0u;
#else
    return (b >= 0 && (u32)(2 * d + 1) < nwords + 128u) ? words32[d] : 0u;
#endif
  };
#if LMC_DEC_RING_AGPR
  // (round 6) The block in flight does not wait in a VGPR the compiler knows about but in AGPR a0, requested and read
  // by inline asm.  Why: a register with a load outstanding costs the compiler an `s_waitcnt vmcnt(0)` where it is read,
  // and on gfx9 vmcnt counts the stores too -- the commit of a block then also waits for the acknowledgement of the
  // rows stored a token or two ago (a decode without its stores runs in 0.66 ms against 0.79).  Vector memory operations
  // complete in the order they were issued, so `s_waitcnt vmcnt(N)` with N = the stores issued since the request
  // (`nst`, counted by the block path; every other path leaves N too small, which only waits longer) is all the block
  // needs.  The compiler never holds a value in a0 (tests/test_host_logic.py holds the kernels at one AGPR).
  u32 nst = 0, nst_req = 0;
  auto block_request = [&](int b) {
    if (b >= 0) {  // (wave-uniform)
      const int d = 64 * b + lane;
      const bool in = (u32)(2 * d + 1) < nwords + 128u;
      const LMC_GLOBAL u32* const ap = words32 + (in ? d : 0);
      asm volatile("v_accvgpr_write_b32 a0, 0" ::: LMC_DEC_A0_CLOBBER);  // lanes behind the stream's end hold 0, as block_load returns
      if (in) asm volatile("global_load_dword a0, %0, off nt" :: "v"(ap) : LMC_DEC_A0_CLOBBER, "memory");
    } else {
      asm volatile("v_accvgpr_write_b32 a0, 0" ::: LMC_DEC_A0_CLOBBER);
    }
    nst_req = nst;
  };
  auto block_take = [&]() -> u32 {
    u32 v;
    const u32 n = nst - nst_req;  // (wave-uniform)
    // (a ladder, because s_waitcnt takes an immediate; a request is normally 8 - 10 stores old when its block is taken)
    if (n >= 16u) asm volatile("s_waitcnt vmcnt(16)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else if (n >= 12u) asm volatile("s_waitcnt vmcnt(12)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else if (n >= 10u) asm volatile("s_waitcnt vmcnt(10)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else if (n >= 8u) asm volatile("s_waitcnt vmcnt(8)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else if (n >= 6u) asm volatile("s_waitcnt vmcnt(6)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else if (n >= 4u) asm volatile("s_waitcnt vmcnt(4)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) :: LMC_DEC_A0_CLOBBER, "memory");
    return v;
  };
#endif
  auto block_commit = [&](int b, u32 v) {
    ring32[((b & 1) << 6) + lane] = v;
    if (!(b & 1) && lane < 32) ring32[128 + lane] = v;  // mirror of slots 0..63
  };
  int top_blk = (int)((nwords + 127u) >> 7) - 1;  // block of the last word (-1: no words at all)
  block_commit(top_blk, block_load(top_blk));
  block_commit(top_blk - 1, block_load(top_blk - 1));
  int e = (int)nwords;             // wave-uniform
  int trig = 128 * top_blk;        // next ring event when e <= trig: the upper block is consumed
#if LMC_DEC_RING_AGPR
  block_request(top_blk - 2);  // the block below the resident two is in flight (in a0)
#else
  u32 pend = block_load(top_blk - 2);  // the block below the resident two is in flight in a register
#endif
  wave_lds_fence();

  // The symbol search is a binary search over the lane's CDF column.  Symbols are 0 .. nsym-1 (nsym = bins - 1),
  // so planes with nsym <= 16 (16 bins: most of them) search entries 0..15 only (TOP = 4), the others 0..31
  // (TOP = 8).  Its first two levels run on three pivots held in registers.
  typedef const __attribute__((address_space(3))) u16* lds_u16p;  // 32-bit LDS pointers: no generic-pointer math
  typedef const __attribute__((address_space(3))) u32* lds_u32p;
  typedef const __attribute__((address_space(3))) float* lds_f32p;
  const bool wide = nsym <= 16u;             // wave-uniform: 16 packed 32-bit entries in quarters, else 33 u16 entries
  const u32 top = wide ? 4u : 8u;
  // the lane's column: u16 entries 128 B apart, or (wide) four 16-byte quarters 1024 B apart
  const u32 col_addr = (u32)(size_t)(lds_u16p)cdfT + ((u32)lane << (wide ? 4 : 1));
  const u32 colB_addr = col_addr + 2048u;    // entry 16 of 33 / quarter 2 of 4
  // first two search levels: three pivots in registers (wide: whole entries, compared against slot << 16 | 0xffff)
  const u32 pA = wide ? *(lds_u32p)(size_t)(col_addr + 1024u) : (u32)*(lds_u16p)(size_t)(col_addr + 1024u);
  const u32 pB = wide ? *(lds_u32p)(size_t)colB_addr : (u32)*(lds_u16p)(size_t)colB_addr;
  const u32 pC = wide ? *(lds_u32p)(size_t)(col_addr + 3072u) : (u32)*(lds_u16p)(size_t)(col_addr + 3072u);
  const u32 lut_addr = (u32)(size_t)(lds_f32p)lut;
  const u32 lut_qbias = col_addr - (lut_addr << 6);  // wide: (q - lut_qbias) >> 6 = &lut[4 * quarter]
  // low bits of a wide search key: all ones below the slot (counts model: one short of it, so that the all-ones
  // entries of symbols behind the last occurring one stay above every key)
  u32 c_ffff = counts_model ? 0x7ffffeu : 0xffffu;
  asm volatile("" : "+v"(c_ffff));
  // narrow: lut[s] with s = (qa - col) >> 7 is at ((qa - col) >> 5) + &lut = (qa - lut_bias) >> 5
  const u32 lut_bias = col_addr - (lut_addr << 5);
  const u64 full_exec = __builtin_amdgcn_read_exec();  // the search narrows exec and restores it from here
  const u32 ring_addr = (u32)(size_t)(lds_u16p)ring;
  const u32 rans_l = counts_model ? LMC_COUNTS_L : LMC_RANS_L;
  const u32 Lv = active ? rans_l : 0u;  // idle lanes never renormalise: x < 0 is never true

  // destination: uniform base (SGPRs) + per-lane 32-bit byte offset
  u32 lane_off = 0;
  u64 ubase = 0;
  if (SYMOUT) {
    ubase = (u64)(a.sym_out + (long long)p * T * a.C);
    lane_off = (u32)c;
  } else {
    const int h = c / a.dst.D, d = c - h * a.dst.D;
    lane_off = (u32)(((long long)h * a.dst.stride_head + d) * 2);
    ubase = uniform_ptr(lmc_plane_base(a.dst, p));
  }
  const int tdst0 = a.dst_tok0 + chunk * a.chunk_tokens;

  // the ring bookkeeping is wave-uniform (SGPRs: its tests are scalar branches); runs when e <= trig
  auto ring_event = [&]() {
    wave_lds_fence();  // every lane's reads of the upper block are done
#if LMC_DEC_RING_AGPR
    block_commit(top_blk - 2, block_take());
    top_blk = __builtin_amdgcn_readfirstlane(top_blk - 1);
    block_request(top_blk - 2);  // requested a whole block -- nine tokens or so -- before it is written to the ring
#else
    block_commit(top_blk - 2, pend);
    top_blk = __builtin_amdgcn_readfirstlane(top_blk - 1);
    pend = block_load(top_blk - 2);  // (round 6) requested a whole block -- nine tokens or so -- before it is written
#endif
    trig = 128 * top_blk;
    wave_lds_fence();
  };
  // The word pop of one token (after the state update): renormalising lanes take this step's words.  Branch
  // free: every lane reads a slot (rank < 64 keeps it inside ring + mirror), the renormalising ones keep it.
  auto decode_pop = [&](u64 mask, bool check) {  // mask = lanes with x < L; check: look after the ring behind this token
    e -= (int)__popcll(mask);
    // scalar: the slot of word e - cnt, ring_addr + 2 * (e % 256) -- as s_and + s_lshl1_add_u32 (left alone the compiler
    // shifts, masks and adds: three scalar instructions of a token step that has fifteen)
    u32 sbase = (u32)e & (DEC_RING_WORDS - 1);
    asm("s_lshl1_add_u32 %0, %0, %1" : "+s"(sbase) : "s"(ring_addr) : "scc");
    // under exec = mask: rank, slot address, word, x = x << 16 | word; then exec is restored.  No branch around it
    // (an empty mask makes the four instructions no-ops) and no select afterwards.
    u32 t;
    asm volatile("s_mov_b64 exec, %[m]\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], %[mlo], 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], %[mhi], %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[sb]\n\t"
                 "ds_read_u16 %[t], %[t]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_perm_b32 %[x], %[t], %[x], %[sel]\n\t"
                 "s_mov_b64 exec, %[full]"
                 : [x] "+v"(x), [t] "=&v"(t)
                 : [m] "s"(mask), [mlo] "s"((u32)mask), [mhi] "s"((u32)(mask >> 32)), [sb] "s"(sbase),
                   [sel] "s"(0x01000504u), [full] "s"(full_exec)
                 : "memory");
    if (check && __builtin_expect(e <= trig, 0)) ring_event();
  };
  // One token: search, state update, word pop.  Returns the symbol as an LDS address: of its dequantisation LUT
  // entry (wide) or of its CDF entry (narrow); `lv` receives the LUT value.
  // A search level on exec:  v_cmpx_le_u32 (exec = lanes whose pivot <= slot) ; moves / adds of those lanes ;
  // s_mov exec, full -- the selects become double-rate moves and the compare feeds no v_cndmask.
#define LMC_SEARCH_STEP(q, piv, slot, STEP_BYTES)                                                   \
  asm("v_cmpx_le_u32_e32 vcc, %1, %2\n\tv_add_u32_e32 %0, %3, %0\n\ts_mov_b64 exec, %4"            \
      : "+v"(q) : "v"(piv), "v"(slot), "i"(STEP_BYTES), "s"(full_exec) : "vcc")
  // `check_tag`: whether the ring is looked after behind this token.  The block path does so behind every SECOND token
  // (round 6): a token takes at most 64 words and a whole block lies below the upper one, so the token after the one
  // that emptied the upper block still finds its words -- half the tests, and every ring event falls just in front of a
  // pair's stores.
  auto decode_token = [&](auto top_tag, auto model_tag, float& lv, auto check_tag) -> u32 {
    constexpr bool CHECK = decltype(check_tag)::value;
    constexpr int TOP = decltype(top_tag)::value;
    constexpr bool WIDE = TOP == 4;
    constexpr bool COUNTS = decltype(model_tag)::value;  // LMC_MODEL_COUNTS: 9-bit slots, start << 23 | freq entries
    constexpr int KSH = COUNTS ? 23 : 16;
    if constexpr (WIDE) {
      // Planes with at most 16 symbols.  Entry i = cdf[i] << 16 | freq[i], so "cdf[i] <= slot" is one unsigned
      // compare of the whole entry with slot << 16 | 0xffff.  Levels 1-2 on the register pivots pick a quarter,
      // ONE ds_read_b128 brings its four entries, levels 3-4 select among them in registers: a single LDS round
      // trip per token in the search, and `r` follows the symbol as the address of its LUT entry.
      if constexpr (COUNTS) {
        // Round 6.  All four levels are exec-predicated moves (levels 1-2 on the register pivots pick the quarter, 3-4
        // pick the entry among the four the ds_read_b128 brought): the replica of this step (tools/probes/decode_step.py)
        // runs 3 - 5 ns per token faster with v_cmpx + v_mov than with the unsigned-minimum selection round 5 used for
        // levels 3-4 (4 v_sub + v_min3 + v_min) -- same instruction count, but moves under a mask cost about half a
        // subtraction.  d = key - entry = slot - start (bits 23 ..), the symbol's LUT address (bits 10 .. 19) and freq
        // (bits 0 .. 9); x = freq * (x >> 9) + (slot - start).  The LUT value is read before the renormalising lanes take
        // over exec (v_cmpx writes exec AND vcc: no scalar exec write, the count and the ranks come from vcc); the word
        // pop follows in the same block, its s_waitcnt covers the LUT read too.  Four scalar instructions stand between
        // the v_cmpx and the v_mbcnt that reads vcc (gfx940 family: two wait states between a VALU write of an SGPR and
        // a VALU read of it).
        u32 sl, q, pm, r, d, f, t, cnt;
        asm("v_lshl_or_b32 %[sl], %[x], 23, %[ffff]\n\t"
            "v_mov_b32_e32 %[q], %[colA]\n\t"
            "v_mov_b32_e32 %[pm], %[pA]\n\t"
            "v_cmpx_le_u32_e32 vcc, %[pB], %[sl]\n\t"
            "v_mov_b32_e32 %[q], %[colB]\n\t"
            "v_mov_b32_e32 %[pm], %[pC]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "v_cmpx_le_u32_e32 vcc, %[pm], %[sl]\n\t"
            "v_add_u32_e32 %[q], 0x400, %[q]\n\t"
            "s_mov_b64 exec, %[full]"
            : [sl] "=&v"(sl), [q] "=&v"(q), [pm] "=&v"(pm)
            : [x] "v"(x), [ffff] "v"(c_ffff), [pA] "v"(pA), [pB] "v"(pB), [pC] "v"(pC), [colA] "v"(col_addr),
              [colB] "v"(colB_addr), [full] "s"(full_exec)
            : "vcc");
        const u32x4_t e4 = *(const __attribute__((address_space(3))) u32x4_t*)(size_t)q;
        u32 e0 = e4.x, e1 = e4.y;
        asm volatile("v_cmpx_le_u32_e32 vcc, %[e2], %[sl]\n\t"
                     "v_mov_b32_e32 %[e0], %[e2]\n\t"
                     "v_mov_b32_e32 %[e1], %[e3]\n\t"
                     "s_mov_b64 exec, %[full]\n\t"
                     "v_cmpx_le_u32_e32 vcc, %[e1], %[sl]\n\t"
                     "v_mov_b32_e32 %[e0], %[e1]\n\t"
                     "s_mov_b64 exec, %[full]\n\t"
                     "v_lshrrev_b32_e32 %[x], 9, %[x]\n\t"
                     "v_sub_u32_e32 %[d], %[sl], %[e0]\n\t"
                     "v_bfe_u32 %[r], %[d], 10, 10\n\t"
                     "ds_read_b32 %[lv], %[r]\n\t"
                     "v_and_b32_e32 %[f], 0x3ff, %[d]\n\t"
                     "v_lshrrev_b32_e32 %[d], 23, %[d]\n\t"
                     "v_mad_u32_u24 %[x], %[x], %[f], %[d]\n\t"
                     "v_cmpx_lt_u32_e32 vcc, %[x], %[L]\n\t"
                     "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
                     "s_sub_i32 %[e], %[e], %[cnt]\n\t"
                     "s_and_b32 %[cnt], %[e], 0xff\n\t"
                     "s_lshl1_add_u32 %[cnt], %[cnt], %[ring]\n\t"
                     "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                     "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                     "v_lshl_add_u32 %[t], %[t], 1, %[cnt]\n\t"
                     "ds_read_u16 %[t], %[t]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_perm_b32 %[x], %[t], %[x], %[sel]\n\t"
                     "s_mov_b64 exec, %[full]"
                     : [x] "+v"(x), [e] "+s"(e), [e0] "+v"(e0), [e1] "+v"(e1), [r] "=&v"(r), [lv] "=&v"(lv), [d] "=&v"(d),
                       [f] "=&v"(f), [t] "=&v"(t), [cnt] "=&s"(cnt)
                     : [sl] "v"(sl), [e2] "v"(e4.z), [e3] "v"(e4.w), [L] "v"(Lv), [ring] "s"(ring_addr),
                       [sel] "s"(0x01000504u), [full] "s"(full_exec)
                     : "vcc", "scc", "memory");
        if (CHECK && __builtin_expect(e <= trig, 0)) ring_event();
        return r;
      } else {
      u32 sl, q, r, pm;
      u64 mask;
      // levels 1-2 as one block, both exec-predicated (no back-to-back v_cndmask on one vcc, which the issue probe
      // tools/probes/valu_rates.py shows at a quarter of the normal rate; in this kernel the two forms time the same)
      asm("v_lshl_or_b32 %[sl], %[x], %[ksh], %[ffff]\n\t"
          "v_mov_b32_e32 %[q], %[colA]\n\t"
          "v_mov_b32_e32 %[pm], %[pA]\n\t"
          "v_cmpx_le_u32_e32 vcc, %[pB], %[sl]\n\t"
          "v_mov_b32_e32 %[q], %[colB]\n\t"
          "v_mov_b32_e32 %[pm], %[pC]\n\t"
          "s_mov_b64 exec, %[full]\n\t"
          "v_cmpx_le_u32_e32 vcc, %[pm], %[sl]\n\t"
          "v_add_u32_e32 %[q], 0x400, %[q]\n\t"
          "s_mov_b64 exec, %[full]"
          : [sl] "=&v"(sl), [q] "=&v"(q), [pm] "=&v"(pm)
          : [x] "v"(x), [ffff] "v"(c_ffff), [pA] "v"(pA), [pB] "v"(pB), [pC] "v"(pC), [colA] "v"(col_addr),
            [colB] "v"(colB_addr), [full] "s"(full_exec), [ksh] "n"(KSH)
          : "vcc");
      const u32x4_t e4 = *(const __attribute__((address_space(3))) u32x4_t*)(size_t)q;
      r = (q - lut_qbias) >> 6;  // LUT entry of the quarter's first symbol: quarter * 1024 -> quarter * 16 bytes
      u32 e0 = e4.x, e1 = e4.y, d;
      // levels 3-4 among the quarter's entries, then x = freq * (x >> 16) + (slot - start) (start is the entry's
      // upper half, freq its lower half) and the renormalisation test, as one block: no hazard padding in between
      asm("v_cmpx_le_u32_e32 vcc, %[e2], %[sl]\n\t"
          "v_mov_b32_e32 %[e0], %[e2]\n\t"
          "v_mov_b32_e32 %[e1], %[e3]\n\t"
          "v_add_u32_e32 %[r], 8, %[r]\n\t"
          "s_mov_b64 exec, %[full]\n\t"
          "v_cmpx_le_u32_e32 vcc, %[e1], %[sl]\n\t"
          "v_mov_b32_e32 %[e0], %[e1]\n\t"
          "v_add_u32_e32 %[r], 4, %[r]\n\t"
          "s_mov_b64 exec, %[full]\n\t"
          "v_sub_u32_sdwa %[d], %[x], %[e0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
          "v_mad_u32_u16 %[x], %[x], %[e0], %[d] op_sel:[1,0,0,0]\n\t"
          "v_cmp_lt_u32_e64 %[m], %[x], %[lv]"
          : [e0] "+v"(e0), [e1] "+v"(e1), [r] "+v"(r), [x] "+v"(x), [d] "=&v"(d), [m] "=&s"(mask)
          : [e2] "v"(e4.z), [e3] "v"(e4.w), [sl] "v"(sl), [full] "s"(full_exec), [lv] "v"(Lv)
          : "vcc");
      if (!SYMOUT) lv = *(lds_f32p)(size_t)r;  // issued here: back by the time the word pop below has its word
      return decode_pop(mask, CHECK), r;
      }
    } else {
      constexpr u32 ESTRIDE = 128u;  // bytes between entries of a lane's column
      u32 slot = x & (COUNTS ? 0x1ffu : 0xffffu);
      asm volatile("" : "+v"(slot));  // keep `slot` a plain VGPR: SDWA compares would cost a wait state each
      // q walks the column: q = &cdf[s] for the largest probed s with cdf[s] <= slot.  Two levels on register pivots,
      u32 q, pm;
      asm("v_mov_b32_e32 %[q], %[colA]\n\t"
          "v_mov_b32_e32 %[pm], %[pA]\n\t"
          "v_cmpx_le_u32_e32 vcc, %[pB], %[sl]\n\t"
          "v_mov_b32_e32 %[q], %[colB]\n\t"
          "v_mov_b32_e32 %[pm], %[pC]\n\t"
          "s_mov_b64 exec, %[full]\n\t"
          "v_cmpx_le_u32_e32 vcc, %[pm], %[sl]\n\t"
          "v_add_u32_e32 %[q], 0x400, %[q]\n\t"  // TOP * ESTRIDE
          "s_mov_b64 exec, %[full]"
          : [q] "=&v"(q), [pm] "=&v"(pm)
          : [sl] "v"(slot), [pA] "v"(pA), [pB] "v"(pB), [pC] "v"(pC), [colA] "v"(col_addr), [colB] "v"(colB_addr),
            [full] "s"(full_exec)
          : "vcc");
      // ... one on the column in LDS (step 4), then the five entries q .. q + 4 in ONE round trip: the last two
      // levels, the symbol's start and end are picked among them by exec-predicated moves
      {
        const u32 v = *(lds_u16p)(size_t)(q + 4 * ESTRIDE);
        LMC_SEARCH_STEP(q, v, slot, 4 * ESTRIDE);
      }
      u32 e0 = *(lds_u16p)(size_t)q, e1 = *(lds_u16p)(size_t)(q + ESTRIDE), e2 = *(lds_u16p)(size_t)(q + 2 * ESTRIDE);
      const u32 e3 = *(lds_u16p)(size_t)(q + 3 * ESTRIDE), e4 = *(lds_u16p)(size_t)(q + 4 * ESTRIDE);
      u32 h, f, d;
      u64 mask;
      if constexpr (!COUNTS) {
        asm("v_cmpx_le_u32_e32 vcc, %[e2], %[slot]\n\t"
            "v_mov_b32_e32 %[e0], %[e2]\n\t"
            "v_mov_b32_e32 %[e1], %[e3]\n\t"
            "v_mov_b32_e32 %[e2], %[e4]\n\t"
            "v_add_u32_e32 %[q], 0x100, %[q]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "v_mov_b32_e32 %[h], %[e1]\n\t"
            "v_cmpx_le_u32_e32 vcc, %[e1], %[slot]\n\t"
            "v_mov_b32_e32 %[e0], %[e1]\n\t"
            "v_mov_b32_e32 %[h], %[e2]\n\t"
            "v_add_u32_e32 %[q], 0x80, %[q]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "v_sub_u16_e32 %[f], %[h], %[e0]\n\t"       // entry 32 is 65536 stored as 0: the 16-bit difference is right
            "v_sub_u32_e32 %[d], %[slot], %[e0]\n\t"
            "v_mad_u32_u16 %[x], %[x], %[f], %[d] op_sel:[1,0,0,0]\n\t"
            "v_cmp_lt_u32_e64 %[m], %[x], %[lv]"
            : [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [q] "+v"(q), [x] "+v"(x), [h] "=&v"(h), [f] "=&v"(f),
              [d] "=&v"(d), [m] "=&s"(mask)
            : [e3] "v"(e3), [e4] "v"(e4), [slot] "v"(slot), [full] "s"(full_exec), [lv] "v"(Lv)
            : "vcc");
      } else {
        asm("v_cmpx_le_u32_e32 vcc, %[e2], %[slot]\n\t"
            "v_mov_b32_e32 %[e0], %[e2]\n\t"
            "v_mov_b32_e32 %[e1], %[e3]\n\t"
            "v_mov_b32_e32 %[e2], %[e4]\n\t"
            "v_add_u32_e32 %[q], 0x100, %[q]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "v_mov_b32_e32 %[h], %[e1]\n\t"
            "v_cmpx_le_u32_e32 vcc, %[e1], %[slot]\n\t"
            "v_mov_b32_e32 %[e0], %[e1]\n\t"
            "v_mov_b32_e32 %[h], %[e2]\n\t"
            "v_add_u32_e32 %[q], 0x80, %[q]\n\t"
            "s_mov_b64 exec, %[full]\n\t"
            "v_sub_u16_e32 %[f], %[h], %[e0]\n\t"
            "v_sub_u32_e32 %[d], %[slot], %[e0]\n\t"
            "v_lshrrev_b32_e32 %[x], 9, %[x]\n\t"
            "v_mad_u32_u24 %[x], %[x], %[f], %[d]\n\t"
            "v_cmp_lt_u32_e64 %[m], %[x], %[lv]"
            : [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [q] "+v"(q), [x] "+v"(x), [h] "=&v"(h), [f] "=&v"(f),
              [d] "=&v"(d), [m] "=&s"(mask)
            : [e3] "v"(e3), [e4] "v"(e4), [slot] "v"(slot), [full] "s"(full_exec), [lv] "v"(Lv)
            : "vcc");
      }
      if (!SYMOUT) lv = *(lds_f32p)(size_t)((q - lut_bias) >> 5);
      return decode_pop(mask, CHECK), q;
    }
  };

  const u32 nskip = SYMOUT ? 0u : (tdst0 < 0 ? min(T, (u32)(-tdst0)) : 0u);  // tokens that land below dst token 0
  auto run = [&](auto src_tag, auto top_tag, auto model_tag) {
    constexpr bool SRC_BF16 = decltype(src_tag)::value;
    constexpr bool WIDE = decltype(top_tag)::value == 4;
    for (u32 t = 0; t < nskip; t++) {  // retrieve()'s first-chunk trim: decode, do not store
      float lv_skip;
      (void)decode_token(top_tag, model_tag, lv_skip, BoolTag<true>{});
    }
    // !PAGED: the rows of this stream through a raw buffer descriptor: base = row of the first stored token,
    // soffset (scalar) = one stride_token further each token, voffset = the lane's channel.  The descriptor's range
    // check (on voffset only) drops the stores of idle lanes, whose voffset is out of range: no exec masking.
    const long long row_step = a.dst.stride_token * 2;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(ubase + (u64)((long long)(tdst0 + (int)nskip) * row_step)), (short)0, (int)0xfffffff0u, 0x00020000);
    const u32 voff = active ? lane_off : 0xfffffff8u;
    u32 soff = 0;
    // per-token scales: lane i of `sc` holds the fp32 scale of token t0 + i (64 tokens per block), fetched with
    // one coalesced load a block ahead and handed to all lanes with v_readlane
    auto scale_bits = [&](u32 tb) -> u32 { return (!SYMOUT && tb + lane < T) ? (u32)scl[tb + lane] : 0u; };
    auto scale_f32 = [&](u32 sb) -> float {
      return SRC_BF16 ? __uint_as_float(sb << 16) : (float)__builtin_bit_cast(_Float16, (unsigned short)sb);
    };
    // Round 6, bf16 -> bf16 rows of the counts model: blocks of EIGHT tokens.  The eight scales are one s_load_dwordx4 a
    // block ahead (bf16 pairs: a scalar shift / mask makes each an fp32 SGPR operand of the multiply: no v_readlane), two
    // tokens share one v_cvt_pk_bf16_f32 and leave as buffer_store_short + buffer_store_short_d16_hi, and the trip count,
    // scale index and row offset cost three scalar instructions per eight tokens instead of per token.  Needs the first
    // scale on a dword boundary (T even, or an even plane; an even trim); whatever is left -- the last T % 8 tokens, every
    // other combination -- takes the one-token loop below.
    // PAGED: a vLLM slot mapping fills a block with consecutive tokens, so a block of eight tokens normally lands on eight
    // rows one stride_token apart.  Before the first token the wave works out every token's row (four coalesced loads of
    // slot_mapping in flight together, lane = token), checks each block of eight against its first row (ds_bpermute + one
    // ballot per 64 tokens) and leaves block b's first row in lane b of two registers; if every block is such a run the
    // stream decodes as contiguous rows do, with a descriptor at the block's row (two v_readlane per EIGHT tokens).  One
    // block that is not (slots in any order, a first token in the middle of a group of eight of its block) sends the
    // whole stream through the one-token loop.
    u32 tgen = nskip;
    if constexpr (!SYMOUT && SRC_BF16 && DT_OUT == LMC_DTYPE_BF16 && decltype(model_tag)::value) {
      typedef u32x4_t __attribute__((aligned(4))) u32x4_a4;
      const u64 sc_addr = uniform_ptr(scl) + 2ull * nskip;
      if ((sc_addr & 2ull) == 0ull && nskip + 8u <= T) {
        const u32 nblk = (u32)__builtin_amdgcn_readfirstlane((int)((T - nskip) >> 3));
        bool all_runs = true;
        u32 glo = 0, ghi = 0;  // PAGED: lane b = byte offset of block b's first row from the plane's base
        if constexpr (PAGED) {
          all_runs = row_step > 0 && row_step < (1ll << 28) && nblk <= 32u;
          const int tfirst = tdst0 + (int)nskip;
          u32 slot[4];  // (the low halves: lmc_tok_off reads a slot as 32 bits too)
          const u32* const sm32 = (const u32*)(a.dst.slot_mapping + tfirst);
#pragma unroll
          for (u32 r = 0; r < 4u; r++)
            slot[r] = (a.dst.slot_mapping && 64u * r + (u32)lane < 8u * nblk) ? sm32[2u * (64u * r + (u32)lane)] : 0u;
          const u32 bs = (u32)a.dst.block_size;
          const bool pow2 = (bs & (bs - 1u)) == 0u;  // (every block size vLLM offers: a shift and a mask, no division)
          const u32 bsh = (u32)__builtin_ctz(bs | 0x80000000u);
          const int first = (lane & ~7) << 2, mine = (lane & 7) << 5;
#pragma unroll
          for (u32 r = 0; r < 4u; r++) {
            if (all_runs && 64u * r < 8u * nblk) {  // (a mapping that fails in its first 64 tokens costs one round)
              const u32 sl = slot[r];
              const u32 blk = pow2 ? sl >> bsh : sl / bs, w = sl - blk * bs;
              const long long off = a.dst.slot_mapping ? ((long long)blk * a.dst.stride_block + (long long)w * a.dst.stride_token) * 2
                                                       : (long long)(tfirst + (int)(64u * r) + lane) * row_step;
              const u32 olo = (u32)(unsigned long long)off, ohi = (u32)((unsigned long long)off >> 32);
              const u32 blo = (u32)__builtin_amdgcn_ds_bpermute(first, (int)olo), bhi = (u32)__builtin_amdgcn_ds_bpermute(first, (int)ohi);
              const long long rel = off - (long long)(((u64)bhi << 32) | (u64)blo);
              all_runs = all_runs && __ballot(64u * r + (u32)lane >= 8u * nblk || rel == (long long)(lane & 7) * row_step) == ~0ull;
              const u32 tlo = (u32)__builtin_amdgcn_ds_bpermute(mine, (int)olo), thi = (u32)__builtin_amdgcn_ds_bpermute(mine, (int)ohi);
              if ((u32)(lane >> 3) == r) { glo = tlo; ghi = thi; }
            }
          }
        }
        if (all_runs) {
          const __attribute__((address_space(4))) u32x4_a4* sp = (const __attribute__((address_space(4))) u32x4_a4*)sc_addr;
          u32x4_t cur = sp[0];
          const u64 rbase = ubase + (PAGED ? 0ull : (u64)((long long)(tdst0 + (int)nskip) * row_step));
          u32x4_t desc = {(u32)rbase, (u32)(rbase >> 32) & 0xffffu, 0xfffffff0u, 0x00020000u};
          auto pair = [&](u32 s2) {
            float lva = 0.0f, lvb = 0.0f;
            (void)decode_token(top_tag, model_tag, lva, BoolTag<false>{});
            (void)decode_token(top_tag, model_tag, lvb, BoolTag<true>{});
            const float va = lva * __uint_as_float(s2 << 16), vb = lvb * __uint_as_float(s2 & 0xffff0000u);
            u32 w;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(va), "v"(vb));
#ifdef LMC_EXP_NO_STORE  // (timing experiment: what the stores and the waits behind them cost; output is wrong)
            asm volatile("" :: "v"(w), "v"(voff), "s"(desc), "s"(soff));
#elif defined(LMC_EXP_ONE_STORE)
            asm volatile("buffer_store_short %0, %1, %2, %3 offen nt"
                         :: "v"(w), "v"(voff), "s"(desc), "s"(soff), "s"(soff + (u32)row_step) : "memory");
#else
            asm volatile("buffer_store_short %0, %1, %2, %3 offen nt\n\t"
                         "buffer_store_short_d16_hi %0, %1, %2, %4 offen nt"
                         :: "v"(w), "v"(voff), "s"(desc), "s"(soff), "s"(soff + (u32)row_step) : "memory");
#endif
            soff += 2u * (u32)row_step;
#if LMC_DEC_RING_AGPR && !defined(LMC_EXP_NO_STORE) && !defined(LMC_EXP_ONE_STORE)
            nst += 2u;  // (exactly the vector memory operations issued above: block_take's vmcnt(N) counts on it)
#endif
          };
          for (u32 b = 0; b < nblk; b++) {
            u32x4_t nxt = cur;
            if (b + 1u < nblk) nxt = sp[b + 1u];
            if constexpr (PAGED) {
              const u32 rlo = (u32)__builtin_amdgcn_readlane((int)glo, (int)b), rhi = (u32)__builtin_amdgcn_readlane((int)ghi, (int)b);
              const u64 rb = ubase + (((u64)rhi << 32) | (u64)rlo);
              desc.x = (u32)rb;
              desc.y = (u32)(rb >> 32) & 0xffffu;
              soff = 0;
            }
            pair(cur.x);
            pair(cur.y);
            pair(cur.z);
            pair(cur.w);
            cur = nxt;
          }
          tgen += 8u * nblk;
        }
      }
    }
    u32 sc_next = scale_bits(tgen);
    for (u32 t0 = tgen; t0 < T; t0 += 64) {
      const u32 nt = (u32)__builtin_amdgcn_readfirstlane((int)min(T - t0, 64u));
      const float sc = scale_f32(sc_next);
      sc_next = scale_bits(t0 + nt);
      // PAGED: lane i works out the block row of token t0 + i once (slot_mapping load + division); every token
      // then takes its row with two v_readlane and the store goes through a descriptor based at that row
      long long tok_off2 = 0;
      if (PAGED && !SYMOUT) tok_off2 = (t0 + (u32)lane < T) ? dec_tok_off(a.dst, tdst0 + (int)(t0 + (u32)lane)) * 2 : 0ll;
      auto one_token = [&](u32 i) {
        float lv = 0.0f;
        const u32 sa = decode_token(top_tag, model_tag, lv, BoolTag<true>{});
        if (SYMOUT) {
          const u32 sym = WIDE ? (sa - lut_addr) >> 2 : (sa - col_addr) >> 7;
          if (active) *((LMC_GLOBAL int8_t*)(ubase + (u64)(t0 + i) * a.C) + lane_off) = (int8_t)sym;
        } else {
          const float scale = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), (int)i));
          float val = lv * scale;
          u16 bits;
          if (DT_OUT == LMC_DTYPE_BF16) bits = __builtin_bit_cast(unsigned short, (__bf16)val);  // v_cvt_pk_bf16_f32, RNE
          else {
            // The product is rounded to fp32 FIRST and to fp16 second, as the reference does (x = xq / C * max1 in fp32,
            // then .to(float16): cachegen_decoder.py:20, 190-200).  Left to itself the compiler fuses the two into ONE
            // v_fma_mixlo_f16 -- a single rounding of the exact product -- which differs by one fp16 ulp wherever the
            // fp32-rounded product is an exact tie (0.4 % of the elements at 22 bins; round 5's random-geometry sweep
            // found it, the 16- / 32-bin parity cases never hit a tie).  The empty asm keeps the multiply apart.
            asm volatile("" : "+v"(val));
            bits = (u16)f2fp16(val);
          }
          if (PAGED) {
            const u32 olo = (u32)__builtin_amdgcn_readlane((int)(u32)tok_off2, (int)i);
            const u32 ohi = (u32)__builtin_amdgcn_readlane((int)(u32)((unsigned long long)tok_off2 >> 32), (int)i);
            __amdgpu_buffer_rsrc_t prow = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(ubase + (((u64)ohi << 32) | (u64)olo)), (short)0, (int)0xfffffff0u, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16((short)bits, prow, (int)voff, 0, 2 /* nt */);
          } else {
            __builtin_amdgcn_raw_buffer_store_b16((short)bits, rsrc, (int)voff, (int)soff, 2 /* nt */);
            soff += (u32)row_step;
          }
#if LMC_DEC_RING_AGPR
          nst += 1u;  // (one store per token on either branch; the loads of this loop only make block_take's N smaller than it could be)
#endif
        }
      };
      // (four tokens per trip -- the back edge is a taken scalar branch per token -- measured the same within the box
      // noise in round 5, twice: 0.894 - 0.911 against 0.903 - 0.962 ms, and over eight alternations of the two builds
      // medians of 0.906 against 0.914 while the UNCHANGED encoder of the same two libraries differed by as much;
      // the decoder's VALU pipes are 98 % busy and a scalar branch costs it nothing)
      for (u32 i = 0; i < nt; i++) one_token(i);
    }
  };
  auto run_src = [&](auto src_tag) {
    if (counts_model) {
      if (top == 4u) run(src_tag, IntTag<4>{}, BoolTag<true>{});
      else run(src_tag, IntTag<8>{}, BoolTag<true>{});
    } else {
      if (top == 4u) run(src_tag, IntTag<4>{}, BoolTag<false>{});
      else run(src_tag, IntTag<8>{}, BoolTag<false>{});
    }
  };
  if (src_dtype == (u32)LMC_DTYPE_BF16) run_src(BoolTag<true>{});
  else run_src(BoolTag<false>{});

#ifdef LMC_EXP_TIMELINE
  LMC_DTL(2, (unsigned long long)wall_clock64());
  {
    u32 hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    LMC_DTL(3, ((unsigned long long)nsym << 48) | ((unsigned long long)xcc << 32) | hwid);
  }
#endif
  const bool state_bad = active && x != rans_l;
  if (e != 0 || __ballot(state_bad)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
  }
}
