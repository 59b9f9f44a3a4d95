// encode_modes.hip -- is the fused encode's time a property of WHERE its buffers lie?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/encode_modes tools/probes/encode_modes.hip \
//         -Iinclude -Llmcache_amd/csrc -llmc_hip -Wl,-rpath,'$ORIGIN/../../lmcache_amd/csrc'
//
// Across processes on one box k_encode_fused takes either ~1.03 or ~1.14 ms for the same 16 k context, while the
// two-kernel path does not move.  This probe stays in ONE process and varies one buffer at a time -- a fresh lmc_ctx
// (its symbol workspace, stream scratch, granules), a fresh input KV, a fresh blob arena -- keeping the old ones
// allocated so that the new ones land elsewhere, and prints the time of every combination with the addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "lmc_hip.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(2); } } while (0)
#define LK(x) do { int r__ = (x); if (r__ != 0) { fprintf(stderr, "lmc error %d at %s:%d\n", r__, __FILE__, __LINE__); exit(3); } } while (0)

__global__ void fill(unsigned short* kv, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const float f = (float)((unsigned)(z >> 32) >> 8) * (1.0f / 16777216.0f);
    const unsigned u = __float_as_uint(f);
    kv[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}

int main(int argc, char** argv) {
  const int L = 32, H = 8, D = 128, ctx_tok = 16384, chunk = 256, C = H * D, P = 2 * L, nchunks = ctx_tok / chunk;
  const int ntrial = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 20;
  const long long nelem = (long long)P * ctx_tok * C;
  std::vector<int32_t> bins(P);
  for (int p = 0; p < P; p++) { const int kv = p >= L, l = p - kv * L; bins[p] = !kv ? (l < 10 ? 32 : 16) : (l < 2 ? 32 : 16); }
  const uint64_t stride = (lmc_blob_bound(L, chunk, H, D) + 15) & ~15ull;
  unsigned* status;
  CK(hipHostMalloc((void**)&status, 64, hipHostMallocMapped));
  status[0] = 0;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<unsigned short*> kvs;
  std::vector<unsigned char*> arenas;
  std::vector<lmc_ctx*> ctxs;
  unsigned* sizes;
  CK(hipMalloc(&sizes, 4 * nchunks));
  auto new_kv = [&]() { unsigned short* p; CK(hipMalloc(&p, nelem * 2)); fill<<<4096, 256>>>(p, nelem); CK(hipDeviceSynchronize()); kvs.push_back(p); };
  auto new_arena = [&]() { unsigned char* p; CK(hipMalloc(&p, stride * nchunks)); arenas.push_back(p); };
  auto new_ctx = [&]() { lmc_ctx* c; LK(lmc_ctx_create(0, &c)); ctxs.push_back(c); };
  auto time_it = [&](int ci, int ki, int ai, int path) -> float {
    lmc_kv_layout lay;
    memset(&lay, 0, sizeof lay);
    lay.dtype = LMC_DTYPE_BF16; lay.num_layers = L; lay.num_heads = H; lay.head_size = D; lay.base = kvs[ki];
    lay.stride_layer = 2ll * ctx_tok * C; lay.stride_kv = (long long)ctx_tok * C; lay.stride_token = C; lay.stride_head = D;
    LK(lmc_ctx_set_encode_path(ctxs[ci], path));
    for (int w = 0; w < 3; w++) LK(lmc_encode_chunks(ctxs[ci], &lay, 0, ctx_tok, chunk, bins.data(), arenas[ai], stride, sizes, status, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; r++) LK(lmc_encode_chunks(ctxs[ci], &lay, 0, ctx_tok, chunk, bins.data(), arenas[ai], stride, sizes, status, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  new_kv(); new_arena(); new_ctx();
  printf("base: kv %p arena %p  fused %.4f  two-kernel %.4f  fused %.4f\n", (void*)kvs[0], (void*)arenas[0],
         time_it(0, 0, 0, LMC_ENCODE_PATH_FUSED), time_it(0, 0, 0, LMC_ENCODE_PATH_TWO_KERNELS), time_it(0, 0, 0, LMC_ENCODE_PATH_FUSED));
  for (int t = 1; t <= ntrial; t++) { new_ctx(); printf("fresh ctx %d (workspace elsewhere), kv 0, arena 0: fused %.4f  two-kernel %.4f\n", t, time_it(t, 0, 0, LMC_ENCODE_PATH_FUSED), time_it(t, 0, 0, LMC_ENCODE_PATH_TWO_KERNELS)); }
  for (int t = 1; t <= ntrial; t++) { new_kv(); printf("fresh kv %d at %p, ctx 0, arena 0: fused %.4f  two-kernel %.4f\n", t, (void*)kvs[t], time_it(0, t, 0, LMC_ENCODE_PATH_FUSED), time_it(0, t, 0, LMC_ENCODE_PATH_TWO_KERNELS)); }
  for (int t = 1; t <= ntrial; t++) { new_arena(); printf("fresh arena %d at %p, ctx 0, kv 0: fused %.4f  two-kernel %.4f\n", t, (void*)arenas[t], time_it(0, 0, t, LMC_ENCODE_PATH_FUSED), time_it(0, 0, t, LMC_ENCODE_PATH_TWO_KERNELS)); }
  printf("again base: fused %.4f\n", time_it(0, 0, 0, LMC_ENCODE_PATH_FUSED));
  printf("status %u\n", status[0]);
  return 0;
}
