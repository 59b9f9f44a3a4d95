// counter_calibration.hip -- what FETCH_SIZE / WRITE_SIZE report for the access patterns of the product's kernels.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/counter_calibration tools/probes/counter_calibration.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o cal -- tools/probes/counter_calibration     (and WRITE_SIZE)
//   python tools/counter_calibration.py <fetch.db> <write.db> > profiles/r04_counter_calibration.md
//
// MI355X_MICROARCH.md: FETCH_SIZE is TCC_EA0_RDREQ x 64 B and reports half of a 16 B / lane streaming read; "other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  Every
// kernel below moves exactly BYTES (1 GiB, four times the Infinity Cache: nothing is served on-die) with ONE pattern
// of the encoder / decoder:
//   rd16_nt     16 B / lane non-temporal loads       the raw KV (quantize_oct_fused)
//   rd16        16 B / lane plain loads              k_quantize
//   rd4         4 B / lane loads, 256 B per wave     the symbol workspace (passes 1 and 2), the decoder's stream words
//   rd4_nt      ... non-temporal                     LMC_SYM_LAST_LOAD
//   rd_lds16    global_load_lds_dwordx4              copy_stream16 (two-kernel placement)
//   wr16        16 B / lane stores                   the symbol workspace
//   wr16_nt     16 B / lane non-temporal stores      copy_stream16
//   wr4_nt      4 B / lane non-temporal stores       the coder's 256-byte pieces (counts_code_stream<true>)
//   wr2_buf     2 B / lane raw buffer stores, nt     the decoder's output rows
// The read kernels fold what they load into one word per thread (a value the compiler cannot drop) and write 4 bytes
// per thread at the very end: their WRITE_SIZE is noise; the write kernels read nothing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); return 2; } } while (0)

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))
static const size_t BYTES = 1ull << 30;
static const int WG = 256, NWG = 4096;

__global__ __launch_bounds__(256) void rd16_nt(const u32x4* src, u32* sink, size_t n16) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)NWG * WG) {
    const u32x4 v = __builtin_nontemporal_load((const GLOBAL u32x4*)src + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  sink[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rd16(const u32x4* src, u32* sink, size_t n16) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)NWG * WG) {
    const u32x4 v = ((const GLOBAL u32x4*)src)[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  sink[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rd4(const u32* src, u32* sink, size_t n4) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n4; i += (size_t)NWG * WG) acc ^= ((const GLOBAL u32*)src)[i];
  sink[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rd4_nt(const u32* src, u32* sink, size_t n4) {
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n4; i += (size_t)NWG * WG)
    acc ^= __builtin_nontemporal_load((const GLOBAL u32*)src + i);
  sink[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rd_lds16(const u32x4* src, u32* sink, size_t n16) {
  typedef __attribute__((address_space(1))) const void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  __shared__ __attribute__((aligned(16))) u32 stage[4 * 256];  // 1 KiB per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u32 acc = 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)NWG * WG) {
    __builtin_amdgcn_global_load_lds((gptr)(src + i), (lptr)(stage + 256 * wave), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= stage[256 * wave + 4 * lane];
    __builtin_amdgcn_wave_barrier();
  }
  sink[(size_t)blockIdx.x * WG + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void wr16(u32x4* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)NWG * WG) {
    const u32 v = (u32)i;
    ((GLOBAL u32x4*)dst)[i] = u32x4{v, v + 1, v + 2, v + 3};
  }
}
__global__ __launch_bounds__(256) void wr16_nt(u32x4* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)NWG * WG) {
    const u32 v = (u32)i;
    __builtin_nontemporal_store(u32x4{v, v + 1, v + 2, v + 3}, (GLOBAL u32x4*)dst + i);
  }
}
__global__ __launch_bounds__(256) void wr4_nt(u32* dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n4; i += (size_t)NWG * WG)
    __builtin_nontemporal_store((u32)i, (GLOBAL u32*)dst + i);
}
// the decoder's output: a wave stores one 128-byte piece (2 B / lane) of a row, rows 2 KiB apart (1024 channels), nt
__global__ __launch_bounds__(256) void wr2_buf(unsigned short* dst, size_t n2) {
  const size_t wave = ((size_t)blockIdx.x * WG + threadIdx.x) >> 6, nwaves = (size_t)NWG * WG / 64;
  const int lane = threadIdx.x & 63;
  // wave w writes group (w % 16) of every row it owns: rows w / 16, w / 16 + nwaves / 16, ...
  const size_t rows = n2 / 1024;
  for (size_t r = wave / 16; r < rows; r += nwaves / 16) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + r * 1024), (short)0, (int)0xfffffff0u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b16((short)(r + lane), rs, (int)(2 * (64 * (wave % 16) + lane)), 0, 2);
  }
}

int main() {
  void *a, *sink;
  CK(hipMalloc(&a, BYTES));
  CK(hipMalloc(&sink, (size_t)NWG * WG * 4));
  CK(hipMemset(a, 1, BYTES));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(rd16_nt, dim3(NWG), dim3(WG), 0, 0, (const u32x4*)a, (u32*)sink, BYTES / 16);
    hipLaunchKernelGGL(rd16, dim3(NWG), dim3(WG), 0, 0, (const u32x4*)a, (u32*)sink, BYTES / 16);
    hipLaunchKernelGGL(rd4, dim3(NWG), dim3(WG), 0, 0, (const u32*)a, (u32*)sink, BYTES / 4);
    hipLaunchKernelGGL(rd4_nt, dim3(NWG), dim3(WG), 0, 0, (const u32*)a, (u32*)sink, BYTES / 4);
    hipLaunchKernelGGL(rd_lds16, dim3(NWG), dim3(WG), 0, 0, (const u32x4*)a, (u32*)sink, BYTES / 16);
    hipLaunchKernelGGL(wr16, dim3(NWG), dim3(WG), 0, 0, (u32x4*)a, BYTES / 16);
    hipLaunchKernelGGL(wr16_nt, dim3(NWG), dim3(WG), 0, 0, (u32x4*)a, BYTES / 16);
    hipLaunchKernelGGL(wr4_nt, dim3(NWG), dim3(WG), 0, 0, (u32*)a, BYTES / 4);
    hipLaunchKernelGGL(wr2_buf, dim3(NWG), dim3(WG), 0, 0, (unsigned short*)a, BYTES / 2);
    CK(hipDeviceSynchronize());
  }
  CK(hipGetLastError());
  printf("counter_calibration: 9 kernels x 3 rounds, %zu bytes each\n", BYTES);
  return 0;
}
