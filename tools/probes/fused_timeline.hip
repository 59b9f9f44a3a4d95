// fused_timeline.hip -- phase time stamps of every work item of one k_encode_fused launch (a library built with
// -DLMC_EXP_TIMELINE: tools/build_variants.sh tl "-DLMC_EXP_TIMELINE=1").  Writes gpurun_out/fused_timeline.bin:
// [items][8] u64 = {start, phase A done, pass 1 done, look-back done, wave 0 coded, all coded, hw id, -} (100 MHz ticks).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/fused_timeline tools/probes/fused_timeline.hip -Iinclude -ldl
//   tools/probes/fused_timeline build_alt/tl/liblmc_hip.so [ctx_tokens] [out.bin]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
__global__ void fill(unsigned short* kv, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned h = hash32((unsigned long long)i);
    float f = (float)(h >> 8) * (1.0f / 16777216.0f);
    unsigned u = __float_as_uint(f);
    kv[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}
static int plane_bins(int p, int L) { const int kv = p >= L, l = p - kv * L; return !kv ? (l < 10 ? 32 : 16) : (l < 2 ? 32 : 16); }
int main(int argc, char** argv) {
  const int L = 32, H = 8, D = 128, chunk = 256, C = H * D, P = 2 * L;
  const int ctx_tok = argc > 2 ? atoi(argv[2]) : 16384;
  const char* outp = argc > 3 ? argv[3] : "gpurun_out/fused_timeline.bin";
  const int nchunks = ctx_tok / chunk;
  const long long nelem = (long long)P * ctx_tok * C;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  auto ctx_create = (decltype(&lmc_ctx_create))dlsym(h, "lmc_ctx_create");
  auto encode = (decltype(&lmc_encode_chunks))dlsym(h, "lmc_encode_chunks");
  auto set_path = (decltype(&lmc_ctx_set_encode_path))dlsym(h, "lmc_ctx_set_encode_path");
  auto timeline = (int (*)(void*, size_t))dlsym(h, "lmc_debug_fused_timeline");
  if (!timeline) { fprintf(stderr, "not a -DLMC_EXP_TIMELINE build\n"); return 1; }
  lmc_ctx* ctx;
  LK(ctx_create(0, &ctx));
  LK(set_path(ctx, LMC_ENCODE_PATH_FUSED));
  unsigned short* kv;
  CK(hipMalloc(&kv, nelem * 2));
  fill<<<4096, 256>>>(kv, nelem);
  std::vector<int32_t> bins(P);
  for (int p = 0; p < P; p++) bins[p] = plane_bins(p, L);
  lmc_kv_layout lay;
  memset(&lay, 0, sizeof lay);
  lay.dtype = LMC_DTYPE_BF16; lay.num_layers = L; lay.num_heads = H; lay.head_size = D; lay.base = kv;
  lay.stride_layer = 2ll * ctx_tok * C; lay.stride_kv = (long long)ctx_tok * C; lay.stride_token = C; lay.stride_head = D;
  const uint64_t stride = (lmc_blob_bound(L, chunk, H, D) + 15) & ~15ull;
  unsigned char* blob; unsigned* sizes; unsigned* status;
  CK(hipMalloc(&blob, stride * nchunks));
  CK(hipMalloc(&sizes, 4 * nchunks));
  CK(hipHostMalloc((void**)&status, 64, hipHostMallocMapped));
  memset(status, 0, 64);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 30; i++) LK(encode(ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
  CK(hipEventRecord(e0, s));
  LK(encode(ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int items = nchunks * P;
  std::vector<unsigned long long> tl((size_t)items * 8);
  if (timeline(tl.data(), tl.size() * 8)) { fprintf(stderr, "timeline copy failed\n"); return 1; }
  FILE* f = fopen(outp, "wb");
  fwrite(tl.data(), 8, tl.size(), f);
  fclose(f);
  printf("%d items, last launch %.4f ms, status %u -> %s\n", items, ms, status[0], outp);
  auto decode = (decltype(&lmc_decode_chunks))dlsym(h, "lmc_decode_chunks");
  auto dtimeline = (int (*)(void*, size_t))dlsym(h, "lmc_debug_decode_timeline");
  if (decode && dtimeline) {
    unsigned short* out;
    CK(hipMalloc(&out, nelem * 2));
    lmc_kv_layout dl = lay;
    dl.base = out;
    for (int i = 0; i < 10; i++) LK(decode(ctx, blob, stride, nchunks, &dl, 0, chunk, status + 1, s));
    CK(hipEventRecord(e0, s));
    LK(decode(ctx, blob, stride, nchunks, &dl, 0, chunk, status + 1, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t nw = (size_t)nchunks * P * ((C + 63) / 64);
    std::vector<unsigned long long> dt((nw < 65536 ? nw : 65536) * 4);
    if (dtimeline(dt.data(), dt.size() * 8)) { fprintf(stderr, "decode timeline copy failed\n"); return 1; }
    std::string dp = std::string(outp) + ".decode";
    f = fopen(dp.c_str(), "wb");
    fwrite(dt.data(), 8, dt.size(), f);
    fclose(f);
    printf("%zu decode waves, last launch %.4f ms, status %u -> %s\n", dt.size() / 4, ms, status[1], dp.c_str());
  }
  return 0;
}
