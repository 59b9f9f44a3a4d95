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


// ---- pack: the blobs of one store call, transposed plane-major on their way to the pinned arena (lmc_format.h) ---
//
// k_pack_scan   one workgroup: the size of every (plane, chunk) segment of planes [p_begin, p_end) from the blobs' stream
//               directories, their exclusive prefix sums in table order on top of the running total of the parts in
//               front -> the pack's offset table (in the pack, and a device copy for the copy kernel); the LAST part
//               also writes the pack header with the total size.  (Round 6: a store whose encode is launched in plane
//               ranges -- lmc_store_pack_parts -- packs each range as soon as it is coded, so that its bytes can leave
//               over PCIe while the later planes are still being encoded.  Pack format v3 orders the segments by
//               PLANE, K planes then V planes, which is the order the encoder finishes them in.)
// k_pack_copy   a fixed number of workgroups walk the part's segments (the last part: the static slots as well) and
//               write them into the pack region (16-byte accesses, four loads in flight per thread).
struct PackArgs {
  const u8* blobs;              // device arena, blob i at blobs + i * stride (what lmc_encode_chunks wrote)
  long long stride;
  const u32* sizes_d;           // [n] blob sizes (0: the chunk's encode failed); read by the LAST part only
  int n, L, G;
  u8* host;                     // where the pack goes: pinned device-mapped host memory, or device memory
  unsigned long long cap;
  unsigned long long* table_d;  // device copy of the offset table, [2 L n + 1], + [2 L n + 1] = the running total of the parts so far
  lmc_pack_header hdr;          // filled in by the host but for total_bytes
  u32* status;
  int p_begin, p_end, last;     // this part: planes [p_begin, p_end) of every chunk; last: p_end == 2 L
  unsigned long long* part_h;   // NULL, or two words for the caller: {offset of the part in the streams region, its bytes}
};

// Bytes of segment idx = (plane, chunk) -- the streams of one plane of one chunk: from the beginning of the plane's
// first stream to the beginning of the next plane's (the end of the streams section for the last plane) -- and where
// it begins in the blob.  Every header word is checked before it is used as an offset.
// check: the blob's header and size word are there -- they are written by the chunk's LAST work item, so only the last
// part of a store can hold them against the geometry; a chunk whose encode did not finish (size word 0) or whose header
// is not a v6 header of this geometry gives 0 bytes, and k_pack_scan fails the pack.  The earlier parts (!check) go by
// the stream directory alone: the caller hands them planes whose SUCCESSOR has been coded too (the end of a plane is the
// `beg` its successor's first stream wrote), of full chunks only (lmc_api.hip splits a store only when it has no ragged
// chunk: every blob then has the layout of a chunk_tokens-token blob).
__device__ __forceinline__ u32 pack_seg_bytes(const PackArgs& a, int idx, u32* begin, bool check) {
  const int p = idx / a.n, chunk = idx - p * a.n;
  const int P = 2 * a.L;
  const u8* blob = a.blobs + (long long)chunk * a.stride;
  const u32* hd = reinterpret_cast<const u32*>(blob);
  if (begin) *begin = 0u;
  BlobOff bo = lmc_blob_off((u32)P, (u32)a.hdr.chunk_tokens, (u32)a.G);
  u32 stream_bytes = (u32)min((unsigned long long)a.stride - bo.streams, 0xfffffff0ull);
  if (check) {
    if (a.sizes_d[chunk] == 0u || hd[0] != LMC_BLOB_MAGIC || (hd[1] & 0xffffu) != LMC_BLOB_VERSION || hd[8] != (u32)P ||
        hd[9] != (u32)a.G)
      return 0u;
    bo = lmc_blob_off((u32)P, hd[4], (u32)a.G);  // the chunk's own length (a ragged last chunk is shorter)
    if (hd[14] != bo.gdir || hd[15] != bo.streams || (unsigned long long)bo.streams + hd[16] > (unsigned long long)a.stride) return 0u;
    stream_bytes = hd[16];
  } else if (p + 1 >= P) {
    return 0u;  // (the last plane ends with the section: the last part's)
  }
  const u32* gdir = reinterpret_cast<const u32*>(blob + bo.gdir);
  const u32 s = gdir[2 * (p * a.G)];
  const u32 e = p + 1 < P ? gdir[2 * ((p + 1) * a.G)] : stream_bytes;
  if (begin) *begin = bo.streams + s;
  return e >= s && e <= stream_bytes && !(s & 15u) && !(e & 15u) ? e - s : 0u;
}

// One part of a pack: planes [p_begin, p_end).  table_d[N + 1] carries the running total from part to part (the host
// zeroes it in front of the first part); table_d[N] == ~0 says "the pack has failed": later parts do nothing.
__global__ __launch_bounds__(256) void k_pack_scan(PackArgs a) {
  __shared__ unsigned long long sums[256];
  const int N = 2 * a.L * a.n;
  const int s0 = a.p_begin * a.n, M = (a.p_end - a.p_begin) * a.n, K = (M + 255) / 256;
  const int t = (int)threadIdx.x, i0 = s0 + t * K, i1 = min(s0 + M, i0 + K);
  const bool check = a.last != 0;
  const bool failed = a.p_begin > 0 && a.table_d[N] == ~0ull;
  unsigned long long mine = 0;
  int empty = 0;  // a segment of no bytes: its chunk's encode did not finish, or its header does not check out
  for (int i = i0; i < i1; i++) {
    const u32 b = pack_seg_bytes(a, i, nullptr, check);
    empty |= b == 0u;
    mine += b;
  }
  sums[t] = mine;
  const bool bad = __syncthreads_or(empty) != 0 || failed;
  unsigned long long off = 0, total = 0;
  for (int j = 0; j < 256; j++) {  // 256 adds per thread: one launch per part
    off += j < t ? sums[j] : 0ull;
    total += sums[j];
  }
  const unsigned long long base = a.p_begin > 0 ? a.table_d[N + 1] : 0ull;
  const bool fits = !bad && a.hdr.off_streams + base + total <= a.cap;
  unsigned long long* table_h = reinterpret_cast<unsigned long long*>(a.host + a.hdr.off_table);
  off += base;
  for (int i = i0; i < i1; i++) {
    a.table_d[i] = off;
    if (fits) table_h[i] = off;
    off += pack_seg_bytes(a, i, nullptr, check);
  }
  __syncthreads();  // (every thread has read table_d[N + 1] before thread 0 moves it on)
  if (t == 0) {
    a.table_d[N + 1] = base + total;
    if (!fits) a.table_d[N] = ~0ull;  // all ones: nothing more is copied
    else if (a.p_begin == 0) a.table_d[N] = 0ull;  // (whatever an earlier job left there)
    if (a.part_h) {
      a.part_h[0] = base;
      a.part_h[1] = fits ? total : 0ull;
    }
    if (a.last) {
      if (fits) {
        a.table_d[N] = base + total;
        table_h[N] = base + total;
        if (a.hdr.off_static > a.hdr.off_table + 8ull * (unsigned long long)(N + 1)) table_h[N + 1] = 0ull;  // alignment pad
      }
      lmc_pack_header h = a.hdr;
      h.total_bytes = fits ? a.hdr.off_streams + base + total : 0ull;  // 0: not a pack
      if (!fits) h.magic = 0u;
      *reinterpret_cast<lmc_pack_header*>(a.host) = h;
    }
    if (!fits && !failed) atomicOr(a.status, LMC_ST_HOST_ARENA_FULL);
  }
}

__device__ __forceinline__ void pack_copy16(uint4* dst, const uint4* src, u32 n16) {
  const u32 step = 256u;
  u32 i = threadIdx.x;
  for (; i + 3u * step < n16; i += 4u * step) {
    const uint4 v0 = src[i], v1 = src[i + step], v2 = src[i + 2u * step], v3 = src[i + 3u * step];
    dst[i] = v0; dst[i + step] = v1; dst[i + 2u * step] = v2; dst[i + 3u * step] = v3;
  }
  for (; i < n16; i += step) dst[i] = src[i];
}

// grid = (workgroups), 256 threads: workgroup w takes items w, w + gridDim.x, ... of the part's segments (+ the n static
// slots in the last part)
__global__ __launch_bounds__(256) void k_pack_copy(PackArgs a) {
  const int N = 2 * a.L * a.n;
  if (a.table_d[N] == ~0ull) return;
  const int s0 = a.p_begin * a.n, M = (a.p_end - a.p_begin) * a.n;
  const bool check = a.last != 0;
  for (int item = (int)blockIdx.x; item < M + (a.last ? a.n : 0); item += (int)gridDim.x) {
    if (item < M) {
      u32 begin;
      const u32 bytes = pack_seg_bytes(a, s0 + item, &begin, check);
      const int chunk = (s0 + item) % a.n;
      pack_copy16(reinterpret_cast<uint4*>(a.host + a.hdr.off_streams + a.table_d[s0 + item]),
                  reinterpret_cast<const uint4*>(a.blobs + (long long)chunk * a.stride + begin), bytes >> 4);
    } else {
      const int chunk = item - M;
      const u8* blob = a.blobs + (long long)chunk * a.stride;
      const u32 bytes = min((reinterpret_cast<const u32*>(blob)[15] + 15u) & ~15u, a.hdr.static_stride);
      uint4* slot = reinterpret_cast<uint4*>(a.host + a.hdr.off_static + (unsigned long long)chunk * a.hdr.static_stride);
      pack_copy16(slot, reinterpret_cast<const uint4*>(blob), bytes >> 4);
      // a ragged last chunk has shorter static sections than its slot: the rest reads zero (a pack is a function of
      // its blobs, byte for byte)
      for (u32 i = (bytes >> 4) + threadIdx.x; i < (a.hdr.static_stride >> 4); i += 256u) slot[i] = make_uint4(0, 0, 0, 0);
    }
  }
}
