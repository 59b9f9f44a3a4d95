// k_offload.h -- the host-DRAM leg of store / retrieve driven from the GPU.
//
// Replaces LMCLocalBackend.put_blocking / put_nonblocking / get (lmcache/storage_backend/local_backend.py:82-100,
// 128-144): `.to("cpu")` of the chunk with a host wait in front of it (the reference's own comment there:
// "synchronize is harmful"), and `.to("cuda")` of the whole chunk before anything can be decoded.
//
//   k_offload        after an encode job: every blob leaves for a pinned, device-mapped host arena at its EXACT size.
//                    The sizes are only known on the GPU; a hipMemcpyAsync per blob needs the host to read them first
//                    (a host wait per job, or per range of a job).  This kernel reads them where they are: blob i goes
//                    to host_arena + sum over j < i of r16(size_j), the kernel writes offsets and sizes to pinned words
//                    for whoever looks later, and the store call returns without having waited for anything.
// The retrieve leg (lmc_load_chunks, lmc_api.hip) needs no kernel of its own: the blobs lie in pinned host memory, so
// the CPU reads their stream directories and issues one hipMemcpyAsync per contiguous plane run of a layer range --
// a gather KERNEL reading the pinned blobs over PCIe was built first and measured at 41 GB/s against the DMA
// engines' 52 (round 3), i.e. slower than the chunk-major copy it was meant to beat.
// k_offload is a PCIe-bound copy with 16-byte accesses; a few dozen workgroups keep the link busy and leave the CUs
// to the model (46 GB/s: 11.1 ms per 16 k context, against 9.8 ms for sized hipMemcpyAsyncs that cost a host wait).
#pragma once
#include "lmc_device.h"

struct OffloadArgs {
  const u8* blobs;              // device arena, blob i at blobs + i * stride
  long long stride;
  const u32* sizes_d;           // [nchunks] total bytes of every blob (device-accessible)
  int nchunks;                  // chunks of the whole job
  int chunk0;                   // this launch copies chunks chunk0 .. chunk0 + gridDim.y (a job leaves in a few parts)
  u8* host;                     // pinned, device-mapped arena
  unsigned long long cap;       // its capacity
  unsigned long long* offsets_h;  // [nchunks + 1] pinned: blob i at host + offsets_h[i]; [nchunks] = bytes used
  u32* sizes_h;                 // [nchunks] pinned copy of the sizes (may alias sizes_d)
  u32* status;
};

#define LMC_ST_HOST_ARENA_FULL LMC_STATUS_HOST_ARENA_FULL

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* sh) {
  // sum over the 256 threads of a workgroup
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += ((unsigned long long)(u32)__shfl_down((int)(u32)v, off) |
          ((unsigned long long)(u32)__shfl_down((int)(u32)(v >> 32), off) << 32));
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// grid = (slices, nchunks), 256 threads
__global__ __launch_bounds__(256) void k_offload(OffloadArgs a) {
  __shared__ unsigned long long sh[4];
  const int chunk = a.chunk0 + (int)blockIdx.y;
  // where does this blob go?  running sum of the padded sizes of the blobs in front of it
  unsigned long long part = 0;
  for (int j = (int)threadIdx.x; j < chunk; j += 256) part += (unsigned long long)((a.sizes_d[j] + 15u) & ~15u);
  const unsigned long long off = block_sum_u64(part, sh);
  const u32 size = a.sizes_d[chunk];
  const unsigned long long padded = (unsigned long long)((size + 15u) & ~15u);
  const bool fits = size != 0u && (unsigned long long)size <= (unsigned long long)a.stride && off + padded <= a.cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.offsets_h[chunk] = off;
    a.sizes_h[chunk] = fits ? size : 0u;
    if (chunk == a.nchunks - 1) a.offsets_h[a.nchunks] = off + (fits ? padded : 0ull);
    if (!fits) atomicOr(a.status, LMC_ST_HOST_ARENA_FULL);
  }
  if (!fits) return;
  const uint4* src = reinterpret_cast<const uint4*>(a.blobs + (long long)chunk * a.stride);
  uint4* dst = reinterpret_cast<uint4*>(a.host + off);
  const u32 n16 = (u32)(padded >> 4);  // blob sizes are multiples of 16 (lmc_format.h)
  const u32 step = gridDim.x * 256u;
  u32 i = blockIdx.x * 256u + threadIdx.x;
  // four 16-byte loads in flight per thread, stores posted over PCIe
  for (; i + 3u * step < n16; i += 4u * step) {
    const uint4 v0 = src[i], v1 = src[i + step], v2 = src[i + 2u * step], v3 = src[i + 3u * step];
    dst[i] = v0; dst[i + step] = v1; dst[i + 2u * step] = v2; dst[i + 3u * step] = v3;
  }
  for (; i < n16; i += step) dst[i] = src[i];
}

