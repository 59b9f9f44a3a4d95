// row_div.hip -- exhaustive check of a short IEEE-exact form of the quantiser's per-row factor = MAX / max
// (k_quantize.h / k_fused.h: `factor[r] = maxf / sf`, today the compiler's 14-instruction v_div_scale /
// v_rcp / Newton / v_div_fmas / v_div_fixup sequence, executed by every lane for a wave-uniform value).
//
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/probes/row_div tools/probes/row_div.hip && ./tools/probes/row_div
//
// Candidate:  y = v_rcp_f32(sf);  q0 = MAX * y;  r = fma(-sf, q0, MAX);  q = fma(r, y, q0)      (4 instructions)
// compared with the correctly rounded quotient for every 16-bit pattern of sf (bf16 and fp16, finite, > 0)
// and every MAX = 1 .. 15 (bins 4 .. 32).  Rows whose factor is inf / NaN or whose max is inf / NaN take the
// quantiser's special path and need the SAME classification from both forms, so those are compared too.
// Prints the mismatch count per dtype and how many of them lie inside the range test the product applies
// (lmc_device.h: row_div_in_range): that second count has to be 0 before LMC_SHORT_ROW_DIV is switched on.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ inline float h2f(unsigned bits, int dtype) {
  if (dtype == 0) return __uint_as_float(bits << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
}

// the product's range test (lmc_device.h: row_div_in_range)
__device__ inline bool in_range(unsigned bits, int dtype) {
  if (dtype == 0) {
    const unsigned e = (bits >> 7) & 0xffu;
    return e >= 27u && e <= 227u;
  }
  return bits != 0u && bits < 0x7c00u;
}

__global__ void check(int dtype, unsigned long long* bad, unsigned* first) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;  // (max pattern, MAX)
  const unsigned bits = i >> 4, m = i & 15u;
  if (bits >= 0x8000u || m == 0u) return;
  const float sf = h2f(bits, dtype), maxf = (float)m;
  const float ref = maxf / sf;  // IEEE
  const float y = __builtin_amdgcn_rcpf(sf);
  const float q0 = maxf * y;
  const float r = __builtin_fmaf(-sf, q0, maxf);
  const float q = __builtin_fmaf(r, y, q0);
  // the quantiser's classification of the row
  const bool sp_ref = !(__builtin_fabsf(ref) < __builtin_inff()) || !(sf < __builtin_inff());
  const bool sp_q = !(__builtin_fabsf(q) < __builtin_inff()) || !(sf < __builtin_inff());
  const bool same = sp_ref ? sp_q : (!sp_q && __float_as_uint(ref) == __float_as_uint(q));
  if (!same) {
    atomicAdd(bad, 1ull);
    atomicMin(first, i);
    if (in_range(bits, dtype)) atomicAdd(bad + 1, 1ull);  // these would reach the product
  }
}

int main() {
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 16);
  hipMalloc(&first, 4);
  for (int dtype = 0; dtype < 2; dtype++) {
    unsigned long long hb[2] = {0, 0};
    unsigned hf = ~0u;
    hipMemcpy(bad, hb, 16, hipMemcpyHostToDevice);
    hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
    check<<<(0x8000u * 16u + 255u) / 256u, 256>>>(dtype, bad, first);
    hipDeviceSynchronize();
    hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("%s: %llu mismatches of %u, %llu of them inside row_div_in_range", dtype == 0 ? "bf16" : "fp16", hb[0],
           0x8000u * 15u, hb[1]);
    if (hb[0]) printf(" (first: max bits 0x%04x, MAX %u)", hf >> 4, hf & 15u);
    printf("\n");
  }
  return 0;
}
