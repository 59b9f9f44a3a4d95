// transpose64 (lmcache_amd/csrc/k_bits.h) against the definition, over random matrices: hipcc --offload-arch=gfx950 -O3
// -std=c++17 -Ilmcache_amd/csrc -Iinclude tools/probes/bit_transpose.hip -o tools/probes/bit_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "k_bits.h"

__global__ void k(const uint64_t* in, uint64_t* out) {
  const int lane = threadIdx.x & 63;
  const uint64_t v = in[blockIdx.x * 64 + lane];
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  transpose64(lo, hi, lane);
  out[blockIdx.x * 64 + lane] = ((uint64_t)hi << 32) | lo;
}

int main() {
  const int NB = 256;
  uint64_t *h = (uint64_t*)malloc(NB * 64 * 8), *o = (uint64_t*)malloc(NB * 64 * 8), *di, *dout;
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < NB * 64; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = i < 64 ? (1ull << (i % 64)) : i < 128 ? (i & 1 ? ~0ull : 0ull) : s; }
  hipMalloc(&di, NB * 64 * 8); hipMalloc(&dout, NB * 64 * 8);
  hipMemcpy(di, h, NB * 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, NB * 64 * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int b = 0; b < NB; b++)
    for (int kk = 0; kk < 64; kk++) {
      uint64_t want = 0;
      for (int l = 0; l < 64; l++) want |= ((h[b * 64 + l] >> kk) & 1ull) << l;
      if (want != o[b * 64 + kk]) { if (bad < 8) printf("block %d lane %d: got %016llx want %016llx\n", b, kk, (unsigned long long)o[b * 64 + kk], (unsigned long long)want); bad++; }
    }
  printf("bit_transpose: %ld of %d rows differ\n", bad, NB * 64);
  return bad != 0;
}
