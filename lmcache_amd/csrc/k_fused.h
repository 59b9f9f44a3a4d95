// k_fused.h -- one workgroup encodes one (chunk, plane) tile end to end: quantise ->
// histogram -> CDF -> interleaved rANS, with the symbols never leaving the CU
// (SURVEY.md section 8a rows a5-a11 in a single launch).
//
// Replaces k_quantize + k_cdf_encode for C = 512 / 1024 channels and chunks of up to 256
// tokens (Llama-3-8B, Mistral-7B: 8 KV heads x 128), i.e. the reference's _split_kv,
// torch_quant_vectorized x2, torch.cat, torchac_cuda.calculate_cdf x2 and
// torchac_cuda.encode_fast_new (cachegen_encoder.py:40-61, 76-91, 255-260, 278-290).
// Bit-identical output to the unfused kernels (same blob bytes).
//
// Workgroup = C threads = NW waves (one per 64-channel group).  The tile's symbols
// (T x C bytes = 256 KiB) do not fit LDS, so they live in REGISTERS: 64 VGPRs per lane.
//   P1  wave w quantises row quads w, w+NW, ... exactly as k_quantize does (16-byte loads,
//       packed-u16 absmax, xor-shuffle reduce); the four tokens of a channel are byte-packed
//       into one dword kept in sym[i][..]; every symbol is counted into a workgroup-wide LDS
//       histogram [bin pair][perm(channel)] (bank = lane: conflict-free ds_add).
//   P2  thread c builds channel c's CDF (exact integer RNE) into tab[entry][channel] u16,
//       which aliases the dead histogram, and the workgroup writes the CDF section coalesced.
//   P3  for each slab of NW consecutive quads (descending): every wave drops its quad of that
//       slab into an LDS slab [quad][channel]; after a barrier wave g (= group g) reads its
//       64 channels' dwords and codes the slab's tokens, descending, table lookups pipelined
//       one token ahead as in k_cdf_encode.
// LDS: 66*C + 4*NW*C bytes = 130 KiB at C = 1024 (one workgroup, 4 waves per SIMD), 49 KiB at
// C = 512.  HBM traffic = the algorithmic bytes: KV read once; scales, CDF, streams written.
// Measured (MI355X, Llama-3-8B 16k context): 1.59 ms vs 0.58 + 0.89 ms for k_quantize +
// k_cdf_encode -- with one workgroup per CU the memory-latency-bound P1 and the VALU-bound P3 do
// not overlap, and P3 runs at 4 waves per SIMD.  Opt-in (lmc_ctx_set_fused); see DESIGN.md 6.
#pragma once
#include "k_encode.h"
#include "k_quantize.h"

// Compile-time loop: the register array sym[][] must only ever be indexed by constants.
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntTag<I>{});
    static_for<N, I + 1>(f);
  }
}

struct FusedArgs {
  KvAddr src;
  BinsArg bins;
  int tok_begin, tok_end, chunk_tokens, nchunks;
  int P, C, G;
  u8* blobs;
  long long blob_stride;
  u8* scratch;
  u32 cap;
  u32* glen;
  u32* status;
};

template <int NITER, int DT>
__global__ __launch_bounds__(512 * NITER, 4) void k_fused_encode(FusedArgs a) {
  constexpr int C = 512 * NITER;       // channels = threads
  constexpr int NW = C / 64;           // waves = 64-channel groups
  constexpr int QPW = 64 / NW;         // row quads per wave (chunk of <= 256 tokens = 64 quads)
  constexpr int RUNS = NITER * 8;      // channels (= dwords per quad) owned by a lane in P1
  constexpr int C8 = C / 8;
  __shared__ __attribute__((aligned(16))) u32 regA[(33 * C * 2 + 3) / 4 > 16 * C ? (33 * C * 2 + 3) / 4 : 16 * C];
  __shared__ __attribute__((aligned(16))) u32 slab[NW * C];
  u32* hist = regA;                              // [16][C] u32 (two u16 counters per dword)
  u16* tab = reinterpret_cast<u16*>(regA);       // [33][C] u16, written after hist is in registers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;                   // (chunk, plane)
  const int p = tile % a.P, chunk = tile / a.P;
  const int tok0 = a.tok_begin + chunk * a.chunk_tokens;
  const int Tc = min(a.chunk_tokens, a.tok_end - tok0);
  const u32 T = (u32)Tc;
  const BlobOff bo = lmc_blob_off((u32)a.P, T, (u32)C, (u32)NW, (u32)a.bins.rowpre[a.P]);
  u8* blob = a.blobs + (long long)chunk * a.blob_stride;

  for (int i = tid; i < 16 * C; i += C) hist[i] = 0;
  __syncthreads();

  // ---- P1: quantise this wave's row quads into registers, count symbols --------------------
  u32 sym[QPW][RUNS];
  {
    long long coff[NITER];
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      const int c0 = (it * 64 + lane) * 8;
      const int h = c0 / a.src.D, d = c0 - h * a.src.D;
      coff[it] = (long long)h * a.src.stride_head + d;
    }
    const u16* pbase = lmc_plane_base(a.src, p);
    const float maxf = (float)((int)a.bins.b[p] / 2 - 1);
    u16* scale_out = reinterpret_cast<u16*>(blob + bo.scales) + (long long)p * Tc;
    static_for<QPW>([&](auto itag) {
      constexpr int i = decltype(itag)::value;
      const int q = i * NW + wave;
      uint4 v[4][NITER];
      bool tv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int t = q * 4 + r;
        tv[r] = t < Tc;
        const u16* rowp = pbase + (tv[r] ? lmc_tok_off(a.src, tok0 + t) : 0);
#pragma unroll
        for (int it = 0; it < NITER; it++) v[r][it] = tv[r] ? ld_global_u4(rowp + coff[it]) : make_uint4(0, 0, 0, 0);
      }
      u32 mrow[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        u32 m = 0;
#pragma unroll
        for (int it = 0; it < NITER; it++) {
          m = pk_max_u16(m, v[r][it].x & 0x7fff7fffu);
          m = pk_max_u16(m, v[r][it].y & 0x7fff7fffu);
          m = pk_max_u16(m, v[r][it].z & 0x7fff7fffu);
          m = pk_max_u16(m, v[r][it].w & 0x7fff7fffu);
        }
        mrow[r] = max(m & 0xffffu, m >> 16);
      }
      u32 m01 = mrow[0] | (mrow[1] << 16), m23 = mrow[2] | (mrow[3] << 16);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        m01 = pk_max_u16(m01, (u32)__shfl_xor((int)m01, off));
        m23 = pk_max_u16(m23, (u32)__shfl_xor((int)m23, off));
      }
      mrow[0] = m01 & 0xffffu; mrow[1] = m01 >> 16; mrow[2] = m23 & 0xffffu; mrow[3] = m23 >> 16;
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (tv[r]) scale_out[q * 4 + r] = (u16)mrow[r];
      }
      float factor[4];
      bool special[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float sf = h2f_rt(mrow[r], DT);
        factor[r] = maxf / sf;  // IEEE fp32 division
        special[r] = !(__builtin_fabsf(factor[r]) < __builtin_inff()) || !(sf < __builtin_inff());
      }
#pragma unroll
      for (int it = 0; it < NITER; it++) {
        u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // byte r of o[e] = symbol of (token 4q+r, channel run element e)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const u32 w[4] = {v[r][it].x, v[r][it].y, v[r][it].z, v[r][it].w};
          u32 sy[8];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            sy[2 * k] = quant_fast(h_lo<DT>(w[k]), factor[r], maxf);
            sy[2 * k + 1] = quant_fast(h_hi<DT>(w[k]), factor[r], maxf);
          }
          if (__ballot(special[r])) {  // wave-uniform and rare
            if (special[r]) {
#pragma unroll
              for (int k = 0; k < 4; k++) {
                sy[2 * k] = quant_special(h_lo<DT>(w[k]), factor[r], maxf);
                sy[2 * k + 1] = quant_special(h_hi<DT>(w[k]), factor[r], maxf);
              }
            }
          }
          if (tv[r]) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
              o[e] |= sy[e] << (8 * r);
              // perm(channel) = e*C/8 + it*64 + lane: bank = lane
              atomicAdd(&hist[(sy[e] >> 1) * C + e * C8 + it * 64 + lane], __umul24(sy[e] & 1u, 65535u) + 1u);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) sym[i][it * 8 + e] = o[e];
      }
    });
  }
  __syncthreads();

  // ---- P2: CDF of channel `tid` ----------------------------------------------------------------
  {
    u32 hreg[16];
    const int pc = (tid & 7) * C8 + (tid >> 3);  // perm(tid)
#pragma unroll
    for (int i = 0; i < 16; i++) hreg[i] = hist[i * C + pc];
    __syncthreads();  // every column is in registers: tab may overwrite hist
    const u32 magic = (T == 1u) ? 0xffffffffu : (u32)(0x100000000ull / T);
    u32 n = 0;
#pragma unroll
    for (int i = 0; i <= 32; i++) {
      const u32 ci = rne_div_u32(n * LMC_CDF_SCALE, T, magic) + (u32)i;
      tab[i * C + tid] = (u16)ci;
      if (i < 32) n += (hreg[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
    }
    __syncthreads();
    // entries 1..R of every row (R = bins - 2), coalesced
    const u32 R = (u32)a.bins.b[p] - 2u;
    const float rcpR = 1.0f / (float)R;
    u16* dst = reinterpret_cast<u16*>(blob + bo.cdf) + (long long)C * a.bins.rowpre[p];
    for (u32 e = tid; e < (u32)C * R; e += C) {
      u32 cl, s;
      divmod_small(e, R, rcpR, cl, s);
      dst[e] = tab[(s + 1u) * C + cl];
    }
  }

  // ---- P3: interleaved rANS, slab by slab ---------------------------------------------------------
  const long long gid = (long long)tile * NW + wave;
  LMC_GLOBAL u8* const outb = (LMC_GLOBAL u8*)(a.scratch + gid * (long long)a.cap);
  const u16* tabl = tab + wave * 64 + lane;  // this lane's column: entry i at tabl[i * C]
  u32 x = LMC_RANS_L;
  u32 wcur = 0;
  auto code_token = [&](u32 st, u32 f) {
    const u32 xh = x >> 16;
    const bool emit = xh >= f;
    const u64 mask = __ballot(emit);
    if (emit) *(LMC_GLOBAL u16*)(outb + ((wcur + lane_rank(mask)) << 1)) = (u16)x;
    x = emit ? xh : x;
    wcur += (u32)__popcll(mask);
    x = rans_put(x, f, 0x10000u - f, st);
  };
  const bool full = Tc == 256;
  static_for<QPW>([&](auto rtag) {
    constexpr int i = QPW - 1 - decltype(rtag)::value;  // slabs in descending token order
    // every wave contributes its quad of slab i
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      u32* d = slab + wave * C + (it * 64 + lane) * 8;
      *reinterpret_cast<uint4*>(d) = make_uint4(sym[i][it * 8 + 0], sym[i][it * 8 + 1], sym[i][it * 8 + 2], sym[i][it * 8 + 3]);
      *reinterpret_cast<uint4*>(d + 4) = make_uint4(sym[i][it * 8 + 4], sym[i][it * 8 + 5], sym[i][it * 8 + 6], sym[i][it * 8 + 7]);
    }
    __syncthreads();
    const u32* scol = slab + wave * 64 + lane;  // this lane's channel: quad j of the slab at scol[j * C]
    if (full) {
      u32 wj = scol[(NW - 1) * C];
      u32 s0 = wj >> 24;
      u32 lo_n = tabl[s0 * C], hi_n = tabl[s0 * C + C];
#pragma unroll
      for (int j = NW - 1; j >= 0; j--) {
        const u32 wnext = j > 0 ? scol[(j - 1) * C] : 0u;  // next quad's four symbols, one quad ahead
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          const u32 st = lo_n, f = (hi_n - lo_n) & 0xffffu;
          if (j > 0 || k > 0) {
            const u32 sn = k > 0 ? (wj >> (8 * (k - 1))) & 0xffu : wnext >> 24;
            lo_n = tabl[sn * C];
            hi_n = tabl[sn * C + C];
          }
          code_token(st, f);
        }
        wj = wnext;
      }
    } else {
#pragma unroll
      for (int j = NW - 1; j >= 0; j--) {
        const u32 wj = scol[j * C];
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          const int t = 4 * (i * NW + j) + k;
          if (t < Tc) {
            const u32 s = (wj >> (8 * k)) & 0xffu;
            const u32 lo = tabl[s * C], hi = tabl[s * C + C];
            code_token(lo, (hi - lo) & 0xffffu);
          }
        }
      }
    }
    __syncthreads();  // the slab is rewritten in the next round
  });
  // tail: states, pad, length (same as k_cdf_encode)
  u16* out = reinterpret_cast<u16*>(a.scratch + gid * (long long)a.cap);
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
