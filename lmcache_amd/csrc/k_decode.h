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
// One wave = one group stream, lane = channel.  Per token: first search level
// on three register pivots, then three dependent probes of the lane's CDF
// column in LDS ([entry][lane] u16, bank = lane/2: conflict free), state
// update, ballot/mbcnt pop of 16-bit words.  The words come from a 512-word LDS
// ring refilled half a ring ahead of the consumer with coalesced loads, so no
// global-memory latency sits on the per-token dependency chain.  6.1 KiB of
// LDS per wave -> 6 waves per SIMD.
#pragma once
#include "lmc_device.h"

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
};

// per-wave LDS: cdfT [33][64] u16 (4224) | word ring: 64-word mirror prefix + 512 x u16 (1152) | lut 32 x f32 (128)
#define DEC_CDF_BYTES 4224
#define DEC_RING_WORDS 512
#define DEC_RING_BYTES (2 * (64 + DEC_RING_WORDS))
#define DEC_WAVE_BYTES (DEC_CDF_BYTES + DEC_RING_BYTES + 128)


__device__ __forceinline__ u64 uniform_ptr(const void* p) {  // a pointer every lane holds -> SGPR pair
  const u64 v = (u64)p;
  const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
  return ((u64)hi << 32) | (u64)lo;
}

template <bool SYMOUT, int DT_OUT, bool PAGED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_decode(DecodeArgs a) {
  __shared__ __attribute__((aligned(16))) u8 lds_all[4 * DEC_WAVE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform -> SGPRs
  const long long gid = (long long)blockIdx.x * 4 + wave;
  const int n = 2 * a.layer_count * a.G;  // streams of a chunk in this launch
  if (gid >= (long long)a.nchunks * n) return;
  u8* wl = lds_all + wave * DEC_WAVE_BYTES;
  u16* cdfT = reinterpret_cast<u16*>(wl);                           // [33][64]: entry-major, bank = lane/2
  u16* ring = reinterpret_cast<u16*>(wl + DEC_CDF_BYTES) + 64;      // stream words by consumption order; ring[-64..-1] mirrors ring[448..511]
  float* lut = reinterpret_cast<float*>(wl + DEC_CDF_BYTES + DEC_RING_BYTES);  // (q - C) / C

  const int chunk = (int)(gid / n);
  const int r = (int)(gid - (long long)chunk * n);
  const int pidx = r / a.G, g = r - pidx * a.G;
  // K planes of the layer range first, then their V planes (plane p = kv * L + layer)
  const int p = pidx < a.layer_count ? a.layer_begin + pidx : (a.P >> 1) + a.layer_begin + pidx - a.layer_count;
  const int pg = p * a.G + g;
  const u8* blob = a.blob_ptrs ? (const u8*)uniform_ptr(a.blob_ptrs[chunk]) : a.blobs + (long long)chunk * a.blob_stride;
  const u32* hd = reinterpret_cast<const u32*>(blob);
  // header words are the same for every lane: read them through SGPRs so all control flow below is scalar
  auto hdw = [&](int i) { return (u32)__builtin_amdgcn_readfirstlane((int)hd[i]); };
  const u32 T = hdw(4);
  const u32 src_dtype = hdw(2);
  const u32 cdf_rows = hdw(19);
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)a.C, (u32)a.G, cdf_rows);
  if (hdw(0) != LMC_BLOB_MAGIC || (hdw(1) & 0xffffu) != LMC_BLOB_VERSION || hdw(7) != (u32)a.C ||
      hdw(8) != (u32)a.P || hdw(15) != bo.streams || cdf_rows > 31u * (u32)a.P || hdw(20) != dev_count_bytes(T) ||
      // every section offset is a function of the fields checked above; the blob must also fit its slot
      (!SYMOUT && (T > (u32)a.chunk_tokens || (unsigned long long)hdw(17) > (unsigned long long)a.blob_stride))) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_HEADER);
    return;
  }
  const int c = g * 64 + lane;
  const bool active = c < a.C;

  // ---- symbol counts of this group -> the CDF table in LDS, [entry][lane] u16 ----------------------------
  // The blob stores the counts of every channel's symbols 0 .. nsym-1 (nsym = bins - 1), one byte each for
  // T <= 256 (lmc_format.h).  They are fetched coalesced, staged transposed in the table's own LDS, and
  // every lane then turns its column into the CDF with the encoder's integer arithmetic.
  const u32 nsym = min(31u, max(3u, (u32)__builtin_amdgcn_readfirstlane((int)blob[bo.bins + p]) - 1u));
  {
    const u32 rp = (u32)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const u16*>(blob + bo.rowpre)[p]);
    if (rp + nsym > cdf_rows) {  // a corrupt row prefix must not send the count loads outside the section
      if (lane == 0) atomicOr(a.status, LMC_ST_BAD_HEADER);
      return;
    }
    const float rcpR = 1.0f / (float)nsym;
    const u32 total = (u32)min(64, a.C - g * 64) * nsym;
    const long long e00 = (long long)a.C * rp + (long long)g * 64 * nsym;
    const bool bytes1 = dev_count_bytes(T) == 1u;  // wave-uniform
    const u8* src8 = blob + bo.cdf + e00;
    const u16* src16 = reinterpret_cast<const u16*>(blob + bo.cdf) + e00;
#pragma unroll 1
    for (u32 e0 = 0; e0 < total; e0 += 64 * 10) {  // <= 31 sweeps of 64, 10 loads in flight
      u32 v[10];
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const u32 e = e0 + i * 64 + lane;
        v[i] = e < total ? (bytes1 ? (u32)src8[e] : (u32)src16[e]) : 0u;
      }
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const u32 e = e0 + i * 64 + lane;
        u32 cl, sidx;
        divmod_small(e, nsym, rcpR, cl, sidx);
        if (e < total) cdfT[sidx * 64 + cl] = (u16)v[i];
      }
    }
    for (u32 i = nsym; i < 32u; i++) cdfT[i * 64 + lane] = 0;  // symbols that cannot occur
    wave_lds_fence();  // the staging was written transposed
    u32 hreg[16];  // this lane's 32 counts, two per register
#pragma unroll
    for (int i = 0; i < 16; i++) hreg[i] = (u32)cdfT[(2 * i) * 64 + lane] | ((u32)cdfT[(2 * i + 1) * 64 + lane] << 16);
    if (bytes1) {  // a count of 256 was stored as 255: the counts of a channel sum to T
      u32 sum = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) sum += (hreg[i] & 0xffffu) + (hreg[i] >> 16);
      const u32 deficit = T - sum;  // 0 or 1 in a well-formed blob
      if (__ballot(deficit != 0u)) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if ((hreg[i] & 0xffffu) == 255u) hreg[i] += deficit & 0xffffu;
          else if ((hreg[i] >> 16) == 255u) hreg[i] += (deficit & 0xffffu) << 16;
        }
      }
    }
    cdf_column_to_lds(hreg, T, nsym, cdfT, lane);
    if (!active) {  // idle lanes: any strictly increasing column keeps the search in range
#pragma unroll
      for (int i = 0; i < 33; i++) cdfT[i * 64 + lane] = (u16)i;
    }
  }
  // ---- dequantisation LUT; the per-token scales are fetched 64 tokens at a time inside the loop ----
  const u16* scl = reinterpret_cast<const u16*>(blob + bo.scales) + (long long)p * T;
  if (!SYMOUT && lane < 32) {
    const float Cf = (float)((int)blob[bo.bins + p] / 2 - 1);
    const float v = (float)lane - Cf;
    lut[lane] = v / Cf;
  }

  // ---- stream ----------------------------------------------------------------
  const u32* gend = reinterpret_cast<const u32*>(blob + bo.gend);
  const u32 end = (u32)__builtin_amdgcn_readfirstlane((int)gend[pg]);
  const u32 prev_end = pg == 0 ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)gend[pg - 1]);
  const u32 start = (prev_end + 15u) & ~15u;
  const u16* words = reinterpret_cast<const u16*>(blob + bo.streams + start);
  // 64-bit comparisons: a corrupt directory entry must not wrap its way past the bounds
  if (prev_end > 0xfffffff0u || (unsigned long long)end < (unsigned long long)start + 256ull ||
      (unsigned long long)bo.streams + (unsigned long long)end > (unsigned long long)hdw(17)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
    return;
  }
  const u32 nwords = ((end - start) >> 1) - 128u;  // 16-bit words in front of the 64 states
  u32 x = (u32)words[nwords + 2 * lane] | ((u32)words[nwords + 2 * lane + 1] << 16);
  // Word k (k = 0, 1, ... in the order the decoder consumes them) is words[nwords - 1 - k]; it is staged
  // in ring[k % 512].  The ring is refilled half a ring (256 words) at a time: the loads of the next half
  // are ISSUED into registers when at most 384 words are ahead of the consumer and WRITTEN to the ring
  // when at most 256 are (by then the older half is fully consumed) -- ~10 tokens later, so their
  // latency is off the per-token dependency chain.  Reads past the end of a corrupt stream stay inside
  // the ring (index masked, loads guarded); the final state / count check reports them.
  u16 pend[4];
  auto ring_issue = [&](u32 k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32 k = k0 + i * 64 + lane;
      pend[i] = k < nwords ? words[nwords - 1 - k] : (u16)0;
    }
  };
  auto ring_commit = [&](u32 k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) ring[(k0 + i * 64 + lane) & (DEC_RING_WORDS - 1)] = pend[i];
    if (k0 & 256u) ring[lane - 64] = pend[3];  // mirror of ring[448 + lane]: lets the consumer index without a wrap
  };
  ring_issue(0);
  ring_commit(0);
  ring_issue(256);
  ring_commit(256);
  u32 filled = 512;     // words [0, filled) are in the ring
  u32 consumed = 0;     // wave-uniform
  u32 trigger = 128;    // next ring event when consumed reaches this (= filled - 384, then filled - 256)
  u32 pending = 0;      // loads for [filled, filled + 256) are in flight
  wave_lds_fence();

  // The symbol search is a binary search over the lane's CDF column.  Symbols are 0 .. nsym-1 (nsym = bins - 1),
  // so planes with nsym <= 16 (16 bins: most of them) search entries 0..15 only (TOP = 4), the others 0..31
  // (TOP = 8).  Its first two levels run on three pivots held in registers.
  typedef const __attribute__((address_space(3))) u16* lds_u16p;  // 32-bit LDS pointers: no generic-pointer math
  const lds_u16p col = (lds_u16p)cdfT + lane;
  const u32 top = nsym <= 16u ? 4u : 8u;  // wave-uniform
  const lds_u16p colB = col + 2u * top * 64u;
  const u32 pA = col[top * 64u], pB = colB[0], pC = col[3u * top * 64u];
  typedef const __attribute__((address_space(3))) float* lds_f32p;
  const u32 col_addr = (u32)(size_t)col;
  const u32 lut_bias = col_addr - ((u32)(size_t)(lds_f32p)lut << 5);
  const lds_u16p ringl = (lds_u16p)ring;
  const u32 Lv = active ? LMC_RANS_L : 0u;  // idle lanes never renormalise: x < 0 is never true

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

  // One token: search, state update, word pop; returns the LDS address of the symbol's CDF entry.
  auto decode_token = [&](auto top_tag) -> u32 {
    constexpr int TOP = decltype(top_tag)::value;
    u32 slot = x & 0xffffu;
    asm volatile("" : "+v"(slot));  // keep `slot` a plain VGPR: SDWA compares would cost a wait state each
    // q walks the column: q = &cdf[s] for the largest probed s with cdf[s] <= slot
    const bool geB = pB <= slot;
    const u32 pm = geB ? pC : pA;
    lds_u16p q = geB ? colB : col;
    {
      const lds_u16p q2 = q + TOP * 64;
      q = pm <= slot ? q2 : q;
    }
#pragma unroll
    for (int step = TOP / 2; step >= 1; step >>= 1) {
      const lds_u16p q2 = q + step * 64;
      const u32 v = *q2;
      q = v <= slot ? q2 : q;
    }
    const u32 lo = q[0], hi = q[64];  // the symbol's own two entries; entry 32 is 65536 stored as 0
    const u32 qa = (u32)(size_t)q;
    const u32 f = (hi - lo) & 0xffffu;
    x = __umul24(f, x >> 16) + slot - lo;
    const bool need = x < Lv;
    const u64 mask = __ballot(need);
    const u32 cnt = (u32)__popcll(mask);
    // the encoder appended this token's words in ascending lane order; counted from the tail that is
    // descending, so rank r of cnt takes consumption index consumed + cnt - 1 - r
    const int last = (int)((consumed + cnt - 1u) & (DEC_RING_WORDS - 1));  // scalar; last - rank is in [-63, 511]
    if (need) x = (x << 16) | (u32)ringl[last - (int)lane_rank(mask)];
    // the ring bookkeeping is wave-uniform: pin it to SGPRs so its tests are scalar branches
    consumed += cnt;
    if (consumed >= trigger) {
      if (pending == 0u) {
        ring_issue(filled);
        pending = 1u;
        trigger = (u32)__builtin_amdgcn_readfirstlane((int)(filled - 256u));
      } else {
        wave_lds_fence();  // every lane's reads of the older half are done
        ring_commit(filled);
        filled = (u32)__builtin_amdgcn_readfirstlane((int)(filled + 256u));
        pending = 0u;
        trigger = (u32)__builtin_amdgcn_readfirstlane((int)(filled - 384u));
        wave_lds_fence();
      }
    }
    return qa;
  };

  const u32 nskip = SYMOUT ? 0u : (tdst0 < 0 ? min(T, (u32)(-tdst0)) : 0u);  // tokens that land below dst token 0
  auto run = [&](auto src_tag, auto top_tag) {
    constexpr bool SRC_BF16 = decltype(src_tag)::value;
    for (u32 t = 0; t < nskip; t++) (void)decode_token(top_tag);  // retrieve()'s first-chunk trim: decode, do not store
    // !PAGED: the row of token t starts at rowp (uniform), one stride_token further each token
    LMC_GLOBAL u8* rowp = (LMC_GLOBAL u8*)ubase + (long long)(tdst0 + (int)nskip) * a.dst.stride_token * 2;
    const long long row_step = a.dst.stride_token * 2;
    // per-token scales: lane i of `sc` holds the fp32 scale of token t0 + i (64 tokens per block), fetched with
    // one coalesced load a block ahead and handed to all lanes with v_readlane
    auto scale_bits = [&](u32 tb) -> u32 { return (!SYMOUT && tb + lane < T) ? (u32)scl[tb + lane] : 0u; };
    auto scale_f32 = [&](u32 sb) -> float {
      return SRC_BF16 ? __uint_as_float(sb << 16) : (float)__builtin_bit_cast(_Float16, (unsigned short)sb);
    };
    u32 sc_next = scale_bits(nskip);
    u32 off = lane_off;
    for (u32 t0 = nskip; t0 < T; t0 += 64) {
      const u32 t1 = min(T, t0 + 64u);
      const float sc = scale_f32(sc_next);
      sc_next = scale_bits(t1);
      for (u32 t = t0; t < t1; t++) {
        const u32 qa = decode_token(top_tag);
        if (SYMOUT) {
          if (active) *((LMC_GLOBAL int8_t*)(ubase + (u64)t * a.C) + lane_off) = (int8_t)((qa - col_addr) >> 7);
        } else {
          const float scale = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), (int)(t - t0)));
          if (active) {
            // lut[s] with s = (qa - col) / 128: its LDS address is ((qa - col) >> 5) + &lut = (qa - lut_bias) >> 5
            const float val = *(lds_f32p)(size_t)((qa - lut_bias) >> 5) * scale;
            u16 bits;
            if (DT_OUT == LMC_DTYPE_BF16) bits = __builtin_bit_cast(unsigned short, (__bf16)val);  // v_cvt_pk_bf16_f32, RNE
            else bits = (u16)f2fp16(val);
            LMC_GLOBAL u8* const row = PAGED ? (LMC_GLOBAL u8*)ubase + lmc_tok_off(a.dst, tdst0 + (int)t) * 2 : rowp;
            asm volatile("" : "+v"(off));  // keeps the zero-extension next to the store: SGPR base + 32-bit VGPR offset
            __builtin_nontemporal_store(bits, (LMC_GLOBAL u16*)(row + off));  // written once, read by someone else later
          }
          rowp += row_step;
        }
      }
    }
  };
  if (src_dtype == (u32)LMC_DTYPE_BF16) {
    if (top == 4u) run(BoolTag<true>{}, IntTag<4>{});
    else run(BoolTag<true>{}, IntTag<8>{});
  } else {
    if (top == 4u) run(BoolTag<false>{}, IntTag<4>{});
    else run(BoolTag<false>{}, IntTag<8>{});
  }

  const bool state_bad = active && x != LMC_RANS_L;
  if (consumed != nwords || __ballot(state_bad)) {
    if (lane == 0) atomicOr(a.status, LMC_ST_BAD_STREAM);
  }
}
