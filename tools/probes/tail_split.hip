// tail_split.hip -- ONE 16 k request encoded as two concurrent launches: the first chunks by the shipped library on the
// caller's stream, the last `ntail` chunks by a second library (e.g. a -DFUSED_WAVES=16 build: items of half the
// duration) on a second, low-priority stream, joined by events.  Question: does a finer-grained last generation shorten
// the launch's drain (profiles/r06_decoder_and_timelines.md: ~0.1 ms per launch for the first and last generation)?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/tail_split tools/probes/tail_split.hip -Iinclude -ldl
//   tools/probes/tail_split rounds reps main=path/to/liblmc_hip.so tail=path/to/other/liblmc_hip.so
//
// Prints ms per 16 k context for the single launch and for every split (tail chunks 8 / 16 / 24 / 32, tail library =
// main or tail), min / median over the rounds, and diffs the blobs of every form against the single launch's.
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
__global__ void fill(unsigned short* kv, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = hash32((unsigned long long)i);
    float f = (float)(h >> 8) * (1.0f / 16777216.0f);
    unsigned u = __float_as_uint(f);
    kv[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
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
  void* h;
  decltype(&lmc_ctx_create) ctx_create;
  decltype(&lmc_encode_chunks) encode;
  decltype(&lmc_ctx_set_encode_path) set_path;
  lmc_ctx* ctx;
};
static Lib load(const char* path) {
  Lib l;
  l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(1); }
  l.ctx_create = (decltype(l.ctx_create))dlsym(l.h, "lmc_ctx_create");
  l.encode = (decltype(l.encode))dlsym(l.h, "lmc_encode_chunks");
  l.set_path = (decltype(l.set_path))dlsym(l.h, "lmc_ctx_set_encode_path");
  LK(l.ctx_create(0, &l.ctx));
  LK(l.set_path(l.ctx, LMC_ENCODE_PATH_FUSED));
  return l;
}
static int plane_bins(int p, int L) { const int kv = p >= L, l = p - kv * L; return !kv ? (l < 10 ? 32 : 16) : (l < 2 ? 32 : 16); }

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: tail_split rounds reps main=path tail=path\n"); return 1; }
  const int rounds = atoi(argv[1]), reps = atoi(argv[2]);
  Lib lm = load(strchr(argv[3], '=') + 1), lt = load(strchr(argv[4], '=') + 1);
  const int L = 32, H = 8, D = 128, ctx_tok = 16384, chunk = 256, C = H * D, P = 2 * L, nchunks = ctx_tok / chunk;
  const long long nelem = (long long)P * ctx_tok * C;
  unsigned short* kv;
  CK(hipMalloc(&kv, nelem * 2));
  fill<<<4096, 256>>>(kv, nelem);
  std::vector<int32_t> bins(P);
  for (int p = 0; p < P; p++) bins[p] = plane_bins(p, L);
  lmc_kv_layout lay;
  memset(&lay, 0, sizeof lay);
  lay.dtype = LMC_DTYPE_BF16;
  lay.num_layers = L; lay.num_heads = H; lay.head_size = D; lay.base = kv;
  lay.stride_layer = 2ll * ctx_tok * C; lay.stride_kv = (long long)ctx_tok * C; lay.stride_token = C; lay.stride_head = D;
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
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least priority (numerically greatest)
  hipStream_t s1, s2;
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  printf("stream priorities: main %d, tail %d\n", hi, lo);
  hipEvent_t e0, e1, ef, ej;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipMemset(blob0, 0, stride * nchunks));
  LK(lm.encode(lm.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob0, stride, sizes, status, s1));
  CK(hipStreamSynchronize(s1));

  // form 0: one launch; forms 1..: (ntail, tail library)
  struct Form { int ntail; bool other; std::vector<double> ms; };
  std::vector<Form> forms;
  forms.push_back({0, false, {}});
  for (int nt : {8, 16, 24, 32}) { forms.push_back({nt, false, {}}); forms.push_back({nt, true, {}}); }
  auto job = [&](const Form& f) {
    if (f.ntail == 0) {
      LK(lm.encode(lm.ctx, &lay, 0, ctx_tok, chunk, bins.data(), blob, stride, sizes, status, s1));
      return;
    }
    const int nmain = nchunks - f.ntail;
    Lib& t = f.other ? lt : lm;
    CK(hipEventRecord(ef, s1));
    CK(hipStreamWaitEvent(s2, ef, 0));
    LK(lm.encode(lm.ctx, &lay, 0, nmain * chunk, chunk, bins.data(), blob, stride, sizes, status, s1));
    LK(t.encode(t.ctx, &lay, nmain * chunk, ctx_tok, chunk, bins.data(), blob + (size_t)nmain * stride, stride, sizes + nmain, status + 1, s2));
    CK(hipEventRecord(ej, s2));
    CK(hipStreamWaitEvent(s1, ej, 0));
  };
  for (Form& f : forms) {
    CK(hipMemsetAsync(blob, 0, stride * nchunks, s1));
    job(f);
    CK(hipStreamSynchronize(s1));
    *bad = 0;
    diff16<<<2048, 256, 0, s1>>>((const uint4*)blob, (const uint4*)blob0, stride * nchunks / 16, bad);
    CK(hipStreamSynchronize(s1));
    printf("tail %2d chunks by %-4s: blobs differ from the single launch's in %llu 16-byte words; status %u %u\n", f.ntail,
           f.other ? "tail" : "main", *bad, status[0], status[1]);
  }
  for (int r = 0; r < rounds; r++) {
    for (size_t k = 0; k < forms.size(); k++) {
      Form& f = forms[(k + r) % forms.size()];
      float ms;
      for (int w = 0; w < 2; w++) job(f);
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < reps; i++) job(f);
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      f.ms.push_back(ms / reps);
    }
  }
  printf("%-28s %9s %9s   (ms per 16 k context, %d rounds x %d jobs)\n", "form", "min", "median", rounds, reps);
  for (Form& f : forms) {
    std::sort(f.ms.begin(), f.ms.end());
    char name[64];
    if (f.ntail == 0) snprintf(name, sizeof name, "one launch");
    else snprintf(name, sizeof name, "%d + %d chunks, tail by %s", nchunks - f.ntail, f.ntail, f.other ? "tail lib" : "main lib");
    printf("%-28s %9.4f %9.4f\n", name, f.ms[0], f.ms[f.ms.size() / 2]);
  }
  return 0;
}
