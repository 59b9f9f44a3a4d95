// Probe: rounding of v_cvt_pk_u8_f32 and exactness of v_pk_mul_f32 / v_pk_add_f32 on gfx950.
// Build + run:  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off cvt_pk_u8.hip -o cvt_pk_u8 && ./cvt_pk_u8
// Result on MI355X (ROCm 7.2): "0 / 910 differ from RNE+saturate", "0 / 4096 differ from separately rounded scalar".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef float float2_t __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned r;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
  out[i] = r;
}
__global__ void kpk(const float* a, const float* b, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float2_t x = {a[2 * i], a[2 * i + 1]}, f = {b[0], b[0]}, m = {b[1], b[1]}, y, z;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(f));
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(z) : "v"(y), "v"(m));
  o[2 * i] = z.x; o[2 * i + 1] = z.y;
}
int main() {
  const int n = 4096;
  float h[n]; unsigned ho[n];
  int k0 = 0;
  for (int i = 0; i < 300 && k0 < n; i++) { h[k0++] = i * 0.5f; h[k0++] = nextafterf(i * 0.5f, 1e9f); h[k0++] = nextafterf(i * 0.5f, -1e9f); }
  h[k0++] = -0.4f; h[k0++] = -0.6f; h[k0++] = -3.0f; h[k0++] = 255.4f; h[k0++] = 255.5f; h[k0++] = 300.f; h[k0++] = NAN; h[k0++] = INFINITY; h[k0++] = -INFINITY; h[k0++] = 1e30f;
  int used = k0;
  float* d; unsigned* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
  hipMemcpy(d, h, used * 4, hipMemcpyHostToDevice);
  k<<<(used + 255) / 256, 256>>>(d, o, used);
  hipMemcpy(ho, o, used * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < used; i++) {
    float z = h[i];
    float r = rintf(z);
    unsigned want = std::isnan(z) ? 0u : (r < 0 ? 0u : (r > 255 ? 255u : (unsigned)r));
    if (ho[i] != want) { if (bad < 20) printf("z=%.9g got %u want(rne,sat) %u\n", z, ho[i], want); bad++; }
  }
  printf("cvt_pk_u8: %d / %d differ from RNE+saturate\n", bad, used);
  // pk mul/add vs scalar
  float a[n], b[2] = {15.0f / 3.1415926f, 15.0f}, ho2[n];
  for (int i = 0; i < n; i++) a[i] = (float)(i - 2048) * 0.0017321f;
  float *da, *db, *dz; hipMalloc(&da, n * 4); hipMalloc(&db, 8); hipMalloc(&dz, n * 4);
  hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, 8, hipMemcpyHostToDevice);
  kpk<<<n / 2 / 256, 256>>>(da, db, dz, n);
  hipMemcpy(ho2, dz, n * 4, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int i = 0; i < n; i++) { volatile float y = a[i] * b[0]; volatile float z = y + b[1]; if (memcmp((const void*)&z, &ho2[i], 4)) bad2++; }
  printf("pk_mul/pk_add: %d / %d differ from separately rounded scalar\n", bad2, n);
  return 0;
}
