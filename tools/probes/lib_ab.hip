// lib_ab.hip -- same-process A/B of builds of liblmc_hip.so: every library encodes and decodes THE SAME buffers
// (one allocation: no placement difference between the builds), alternating, several rounds.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/lib_ab tools/probes/lib_ab.hip -Iinclude -ldl
//   tools/probes/lib_ab rounds reps dtype(0 bf16 / 1 fp16) name=path/to/liblmc_hip.so [name=path ...]
//
// Prints per library the minimum and the median over the rounds of: fused encode ms, decode ms (HIP events over `reps`
// back-to-back jobs of the Llama-3-8B 16 k context), and checks that every library's blobs equal the first one's.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "lmc_hip.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(2); } } while (0)
#define LK(x) do { int r__ = (x); if (r__ != 0) { fprintf(stderr, "lmc error %d at %s:%d\n", r__, __FILE__, __LINE__); exit(3); } } while (0)

__device__ inline unsigned hash32(unsigned long long i) {
  unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)(z >> 32);
}
__global__ void fill(unsigned short* kv, long long n, int dtype) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = hash32((unsigned long long)i);
    float f = (float)(h >> 8) * (1.0f / 16777216.0f);
    unsigned short b;
    if (dtype == 0) { unsigned u = __float_as_uint(f); b = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
    else b = __builtin_bit_cast(unsigned short, (_Float16)f);
    kv[i] = b;
  }
}
__global__ void diff16(const uint4* a, const uint4* b, unsigned long long n16, unsigned long long* bad) {
  unsigned long long c = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) {
    uint4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) c++;
  }
  if (c) atomicAdd(bad, c);
}

struct Lib {
  std::string name;
  void* h;
  decltype(&lmc_ctx_create) ctx_create;
  decltype(&lmc_encode_chunks) encode;
  decltype(&lmc_decode_chunks) decode;
  decltype(&lmc_ctx_set_encode_path) set_path;
  lmc_ctx* ctx;
  std::vector<double> enc, dec, pdec, bdec, benc, penc;
};

static int plane_bins(int p, int L) { const int kv = p >= L, l = p - kv * L; return !kv ? (l < 10 ? 32 : 16) : (l < 2 ? 32 : 16); }

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: lib_ab rounds reps dtype name=path ...\n"); return 1; }
  const int rounds = atoi(argv[1]), reps = atoi(argv[2]), dtype = atoi(argv[3]);
  const int L = 32, H = 8, D = 128, ctx_tok = 16384, chunk = 256, C = H * D, P = 2 * L, nchunks = ctx_tok / chunk;
  const long long nelem = (long long)P * ctx_tok * C;
  std::vector<Lib> libs;
  for (int i = 4; i < argc; i++) {
    Lib l;
    const char* eq = strchr(argv[i], '=');
    l.name = std::string(argv[i], eq - argv[i]);
    l.h = dlopen(eq + 1, RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", eq + 1, dlerror()); return 1; }
    l.ctx_create = (decltype(l.ctx_create))dlsym(l.h, "lmc_ctx_create");
    l.encode = (decltype(l.encode))dlsym(l.h, "lmc_encode_chunks");
    l.decode = (decltype(l.decode))dlsym(l.h, "lmc_decode_chunks");
    l.set_path = (decltype(l.set_path))dlsym(l.h, "lmc_ctx_set_encode_path");
    LK(l.ctx_create(0, &l.ctx));
    LK(l.set_path(l.ctx, LMC_ENCODE_PATH_FUSED));
    libs.push_back(l);
  }
  unsigned short *kv, *out;
  CK(hipMalloc(&kv, nelem * 2));
  CK(hipMalloc(&out, nelem * 2 + 65536));  // (slack: the LMC_EXP_STORE_* timing builds write past a row)
  fill<<<4096, 256>>>(kv, nelem, dtype);
  std::vector<int32_t> bins(P);
  for (int p = 0; p < P; p++) bins[p] = plane_bins(p, L);
  lmc_kv_layout lay;
  memset(&lay, 0, sizeof lay);
  lay.dtype = dtype == 0 ? LMC_DTYPE_BF16 : LMC_DTYPE_FP16;
  lay.num_layers = L; lay.num_heads = H; lay.head_size = D; lay.base = kv;
  lay.stride_layer = 2ll * ctx_tok * C; lay.stride_kv = (long long)ctx_tok * C; lay.stride_token = C; lay.stride_head = D;
  lmc_kv_layout dl = lay;
  dl.base = out;
  // the same buffer as a paged cache: per plane [blocks][H][16][D] (the north star's NHBD), slots = a fixed permutation
  lmc_kv_layout pl = dl;
  const int bsz = 16;
  pl.stride_block = (long long)H * bsz * D; pl.stride_token = D; pl.stride_head = (long long)bsz * D; pl.block_size = bsz;
  {
    std::vector<long long> sm(ctx_tok);
    for (int i = 0; i < ctx_tok; i++) sm[i] = (long long)(((long long)i * 7919 + 13) % ctx_tok);
    long long* dsm;
    CK(hipMalloc(&dsm, 8 * ctx_tok));
    CK(hipMemcpy(dsm, sm.data(), 8 * ctx_tok, hipMemcpyHostToDevice));
    pl.slot_mapping = (const int64_t*)dsm;
  }
  // ... and with the slot mapping vLLM produces: blocks anywhere (a fixed permutation of the blocks), a block's sixteen
  // tokens in order
  lmc_kv_layout bl = pl;
  {
    const int nb = ctx_tok / bsz;
    std::vector<long long> sm(ctx_tok);
    for (int i = 0; i < ctx_tok; i++) sm[i] = (long long)((((long long)(i / bsz) * 613 + 7) % nb) * bsz + i % bsz);
    long long* dsm;
    CK(hipMalloc(&dsm, 8 * ctx_tok));
    CK(hipMemcpy(dsm, sm.data(), 8 * ctx_tok, hipMemcpyHostToDevice));
    bl.slot_mapping = (const int64_t*)dsm;
  }
  const uint64_t stride = (lmc_blob_bound(L, chunk, H, D) + 15) & ~15ull;
  unsigned char *blob, *blob0;
  unsigned* sizes;
  CK(hipMalloc(&blob, stride * nchunks));
  CK(hipMalloc(&blob0, stride * nchunks));
  CK(hipMalloc(&sizes, 4 * nchunks));
  unsigned* status;
  CK(hipHostMalloc((void**)&status, 64, hipHostMallocMapped));
  memset(status, 0, 64);
  unsigned long long* bad;
  CK(hipHostMalloc((void**)&bad, 8, hipHostMallocMapped));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // reference blobs and decoded KV from the first library
  CK(hipMemset(blob0, 0, stride * nchunks));
  LK(libs[0].encode(libs[0].ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob0, stride, sizes, status, s));
  CK(hipStreamSynchronize(s));
  unsigned short* out0;
  CK(hipMalloc(&out0, nelem * 2));
  { lmc_kv_layout d0 = lay; d0.base = out0; LK(libs[0].decode(libs[0].ctx, blob0, stride, nchunks, &d0, 0, chunk, status + 1, s)); CK(hipStreamSynchronize(s)); }
  for (size_t k = 0; k < libs.size(); k++) {
    Lib& l = libs[k];
    CK(hipMemset(blob, 0, stride * nchunks));
    CK(hipMemset(out, 0xff, nelem * 2));
    LK(l.encode(l.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
    LK(l.decode(l.ctx, blob, stride, nchunks, &dl, 0, chunk, status + 1, s));
    CK(hipStreamSynchronize(s));
    *bad = 0;
    // (the blob arena between the blobs is never written: zero in both)
    diff16<<<2048, 256, 0, s>>>((const uint4*)blob, (const uint4*)blob0, stride * nchunks / 16, bad);
    CK(hipStreamSynchronize(s));
    const unsigned long long bb = *bad;
    *bad = 0;
    diff16<<<2048, 256, 0, s>>>((const uint4*)out, (const uint4*)out0, nelem * 2 / 16, bad);
    CK(hipStreamSynchronize(s));
    printf("%-8s blobs differ from %s's in %llu 16-byte words, decoded KV in %llu; status %u %u\n", l.name.c_str(), libs[0].name.c_str(), bb, *bad, status[0], status[1]);
  }
  unsigned long long drift = 0;
  for (int r = 0; r < rounds; r++) {
    for (size_t kk = 0; kk < libs.size(); kk++) {
      Lib& l = libs[(kk + r) % libs.size()];
      float ms;
      for (int w = 0; w < 2; w++) LK(l.encode(l.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; i++) LK(l.encode(l.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      l.enc.push_back(ms / reps);
      for (int w = 0; w < 2; w++) LK(l.decode(l.ctx, blob, stride, nchunks, &dl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; i++) LK(l.decode(l.ctx, blob, stride, nchunks, &dl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      l.dec.push_back(ms / reps);
      // what the last of those back-to-back decodes left behind equals the first library's reference output (the decoder
      // waits for its word ring with s_waitcnt vmcnt(N > 0): this is where a wrong N would show)
      diff16<<<2048, 256, 0, s>>>((const uint4*)out, (const uint4*)out0, nelem * 2 / 16, bad);
      CK(hipStreamSynchronize(s));
      drift += *bad;
      *bad = 0;
      for (int w = 0; w < 2; w++) LK(l.decode(l.ctx, blob, stride, nchunks, &pl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; i++) LK(l.decode(l.ctx, blob, stride, nchunks, &pl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      l.pdec.push_back(ms / reps);
      for (int w = 0; w < 2; w++) LK(l.decode(l.ctx, blob, stride, nchunks, &bl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; i++) LK(l.decode(l.ctx, blob, stride, nchunks, &bl, 0, chunk, status + 1, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
      l.bdec.push_back(ms / reps);
      // the encode with the paged cache as its SOURCE (store_paged: the gather is part of phase A), both mappings
      for (int which = 0; which < 2; which++) {
        lmc_kv_layout src = which ? pl : bl;
        src.base = kv;
        for (int w = 0; w < 2; w++) LK(l.encode(l.ctx, &src, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) LK(l.encode(l.ctx, &src, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        (which ? l.penc : l.benc).push_back(ms / reps);
      }
      LK(l.encode(l.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));  // the blobs the decode legs read
    }
  }
  printf("%-8s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s  (ms per 16 k context, %d rounds x %d jobs; pdec = decode + scatter into a paged cache, every token at a slot of its own; bdec = the same with vLLM's mapping: blocks anywhere, a block's tokens in order; benc / penc = encode reading the paged cache through those two mappings)\n", "library", "enc min", "enc med", "dec min", "dec med", "pdec min", "pdec med", "bdec min", "bdec med", "benc min", "benc med", "penc min", "penc med", rounds, reps);
  for (Lib& l : libs) {
    std::vector<double>* v[6] = {&l.enc, &l.dec, &l.pdec, &l.bdec, &l.benc, &l.penc};
    printf("%-8s", l.name.c_str());
    for (auto* x : v) {
      std::sort(x->begin(), x->end());
      printf(" %9.4f %9.4f", (*x)[0], (*x)[x->size() / 2]);
    }
    printf("\n");
  }
  printf("decoded KV after every timed batch of decodes: %llu 16-byte words differ from the reference output\n", drift);
  return drift ? 4 : 0;
}
