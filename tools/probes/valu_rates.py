"""Probe: issue cost (cycles per wave-instruction per SIMD) of the VALU / LDS instructions the coder is
built from, on gfx950.  Generates one kernel per instruction (32 independent copies per loop trip, eight
destination registers in rotation), times it with s_memtime at 1 and 4 waves per SIMD.

    python tools/probes/valu_rates.py            # on the GPU box; prints a table, writes gpurun_out/valu_rates.txt

Result on MI355X (ROCm 7.2) is recorded in HISTORY.md ("What bounds what").
"""
import os
import subprocess
import sys

OPS = [
    # name, asm template (d = dest, a/b/c = sources), clobbers vcc?
    ("v_add_u32", "v_add_u32 {d}, {a}, {b}"),
    ("v_and_b32", "v_and_b32 {d}, {a}, {b}"),
    ("v_lshrrev_b32", "v_lshrrev_b32 {d}, 16, {a}"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {a}, 8, 8"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {a}, 9, {b}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {a}, {b}, 1"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {a}, {b}, {c}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {a}, 4, {b}"),
    ("v_add3_u32", "v_add3_u32 {d}, {a}, {b}, {c}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {a}, {b}, {c}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {a}, {b}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {a}, {b}"),
    ("v_mul_hi_u32", "v_mul_hi_u32 {d}, {a}, {b}"),
    ("v_cndmask_b32", "v_cndmask_b32 {d}, {a}, {b}, vcc"),
    ("v_cndmask_b32_e64", "v_cndmask_b32_e64 {d}, {a}, {b}, s[20:21]"),
    ("v_cmp_ge_u32", "v_cmp_ge_u32 vcc, {a}, {b}"),
    ("v_cmp_ge_u32_sgpr", "v_cmp_ge_u32 s[20:21], {a}, {b}"),
    ("v_sub_co_u32", "v_sub_co_u32 {d}, vcc, {a}, {b}"),
    ("v_addc_co_u32", "v_addc_co_u32 {d}, vcc, {a}, 0, vcc"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {a}"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {d}, {a}"),
    ("v_fma_f32", "v_fma_f32 {d}, {a}, {b}, {c}"),
    ("v_mul_f32", "v_mul_f32 {d}, {a}, {b}"),
    ("v_rcp_f32", "v_rcp_f32 {d}, {a}"),
    ("v_pk_mul_f32", "v_pk_mul_f32 {d2}, {a2}, {b2}"),
    ("v_pk_add_f32", "v_pk_add_f32 {d2}, {a2}, {b2}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {d2}, {a2}, {b2}, {c2}"),
    ("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 {d}, s20, {a}"),
    ("v_mbcnt_hi", "v_mbcnt_hi_u32_b32 {d}, s21, {a}"),
    ("v_cvt_f64_u32", "v_cvt_f64_u32 {d2}, {a}"),
    ("v_cvt_u32_f64", "v_cvt_u32_f64 {d}, {a2}"),
    ("v_mul_f64", "v_mul_f64 {d2}, {a2}, {b2}"),
    ("v_fma_f64", "v_fma_f64 {d2}, {a2}, {b2}, {c2}"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("v_mul_u32_u24_sdwa", "v_mul_u32_u24_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"),
    ("v_cndmask_b32_sdwa", "v_cndmask_b32_sdwa {d}, {a}, {b}, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("v_cmp_ge_u32_sdwa", "v_cmp_ge_u32_sdwa vcc, {a}, {b} src0_sel:WORD_1 src1_sel:WORD_0"),
    ("v_lshlrev_b32_sdwa", "v_lshlrev_b32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"),
    ("v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 {d}, {a}, 1, {b}"),
    ("v_pk_max_u16", "v_pk_max_u16 {d}, {a}, {b}"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {a}, {b}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {a}, {b}, {c}"),
    ("v_perm_b32", "v_perm_b32 {d}, {a}, {b}, {c}"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {a}, {b}, 16"),
    ("v_max3_u32", "v_max3_u32 {d}, {a}, {b}, {c}"),
    ("v_mov_b32_dpp", "v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp", "v_add_u32_dpp {d}, {a}, {b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_readlane_b32", "v_readlane_b32 s22, {a}, 5"),
    ("v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 {d}, {a}"),
    ("v_log_f32", "v_log_f32 {d}, {a}"),
    ("v_ldexp_f32", "v_ldexp_f32 {d}, {a}, {b}"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {a}, {b}, {c}"),
    ("v_sad_u32", "v_sad_u32 {d}, {a}, {b}, {c}"),
    ("v_mad_u64_u32", "v_mad_u64_u32 {d2}, vcc, {a}, {b}, {c2}"),
    ("v_add_u32_e64", "v_add_u32_e64 {d}, {a}, {b}"),
    ("v_xor_b32", "v_xor_b32 {d}, {a}, {b}"),
    ("v_sub_u32", "v_sub_u32 {d}, {a}, {b}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {d}, 3, {a}"),
    ("v_max_u32", "v_max_u32 {d}, {a}, {b}"),
    ("v_add_f32", "v_add_f32 {d}, {a}, {b}"),
    ("v_mov_b32", "v_mov_b32 {d}, {a}"),
    ("v_or_b32", "v_or_b32 {d}, {a}, {b}"),
    ("v_cvt_f32_f16", "v_cvt_f32_f16 {d}, {a}"),
    ("v_mul_i32_i24", "v_mul_i32_i24 {d}, {a}, {b}"),
    ("v_mad_i32_i24", "v_mad_i32_i24 {d}, {a}, {b}, {c}"),
    ("v_add_co_u32", "v_add_co_u32 {d}, vcc, {a}, {b}"),
    ("v_cmp_lt_u32_e64s", "v_cmp_lt_u32_e64 s[28:29], {a}, {b}"),
    ("s_nop_only", "s_nop 0"),
    # groups: cost shown is per QUARTER group (8 groups per 32 slots)
    ("G cmp_vcc+nop1+cnd_e32", "v_cmp_lt_u32 vcc, {a}, {b}; s_nop 1; v_cndmask_b32 {d}, {a}, {b}, vcc"),
    ("G cmp_sgpr+nop1+cnd_e64", "v_cmp_lt_u32_e64 s[28:29], {a}, {b}; s_nop 1; v_cndmask_b32_e64 {d}, {a}, {b}, s[28:29]"),
    ("G cmp_vcc+nop1+cnd_e64vcc", "v_cmp_lt_u32 vcc, {a}, {b}; s_nop 1; v_cndmask_b32_e64 {d}, {a}, {b}, vcc"),
    ("G cmp_vcc+addc", "v_cmp_lt_u32 vcc, {a}, {b}; s_nop 1; v_addc_co_u32 {d}, vcc, {a}, 0, vcc"),
    ("G cmp+saveexec+mov", "v_cmp_lt_u32 vcc, {a}, {b}; s_and_saveexec_b64 s[28:29], vcc; v_mov_b32 {d}, {a}; s_or_b64 exec, exec, s[28:29]"),
    ("G sub+ashr+and_or(3)", "v_sub_u32 {d}, {a}, {b}; v_ashrrev_i32 {d}, 31, {d}; v_and_or_b32 {d}, {d}, {b}, {a}"),
    ("G 2xcnd_e32_samevcc", "v_cmp_lt_u32 vcc, {a}, {b}; s_nop 1; v_cndmask_b32 {d}, {a}, {b}, vcc; v_cndmask_b32 {e}, {b}, {a}, vcc"),
    ("G cnd_e32,add,add,add", "v_cndmask_b32 {d}, {a}, {b}, vcc; v_add_u32 {e}, {a}, {b}; v_add_u32 {e}, {a}, {b}; v_add_u32 {e}, {a}, {b}"),
    ("G cnd_e32,fma x3", "v_cndmask_b32 {d}, {a}, {b}, vcc; v_fma_f32 {e}, {a}, {b}, {c}; v_fma_f32 {e}, {a}, {b}, {c}; v_fma_f32 {e}, {a}, {b}, {c}"),
    # exec-predicated selection (k_decode / k_cdf_encode round 2) against the compare + select form
    ("G cmpx+mov+mov+add+smov", "v_cmpx_le_u32 vcc, {a}, {b}; v_mov_b32 {d}, {a}; v_mov_b32 {e}, {b}; v_add_u32 {d}, 8, {d}; s_mov_b64 exec, s[30:31]"),
    ("G cmp+nop1+cnd+cnd+addc", "v_cmp_le_u32 vcc, {a}, {b}; s_nop 1; v_cndmask_b32 {d}, {a}, {b}, vcc; v_cndmask_b32 {e}, {b}, {a}, vcc; v_addc_co_u32 {d}, vcc, {d}, {d}, vcc"),
    ("G cmpx+add+smov", "v_cmpx_le_u32 vcc, {a}, {b}; v_add_u32 {d}, 8, {d}; s_mov_b64 exec, s[30:31]"),
    ("G cmp+nop1+cnd", "v_cmp_le_u32 vcc, {a}, {b}; s_nop 1; v_cndmask_b32 {d}, {a}, {b}, vcc"),
    ("G mov+mov+mov+mov", "v_mov_b32 {d}, {a}; v_mov_b32 {e}, {b}; v_mov_b32 {d}, {b}; v_mov_b32 {e}, {a}"),
    ("G dep add chain x4", "v_add_u32 {d}, {d}, {a}; v_add_u32 {d}, {d}, {b}; v_add_u32 {d}, {d}, {a}; v_add_u32 {d}, {d}, {b}"),
    ("G dep mad chain x4", "v_mad_u32_u24 {d}, {d}, {a}, {b}; v_mad_u32_u24 {d}, {d}, {a}, {b}; v_mad_u32_u24 {d}, {d}, {a}, {b}; v_mad_u32_u24 {d}, {d}, {a}, {b}"),
    ("ds_read_b32", "ds_read_b32 {d}, {l}"),
    ("ds_read_b64", "ds_read_b64 {d2}, {l8}"),
    ("ds_read_u16", "ds_read_u16 {d}, {l}"),
    ("ds_write_b16", "ds_write_b16 {l}, {a}"),
    ("ds_write_b32", "ds_write_b32 {l}, {a}"),
    ("ds_add_u32", "ds_add_u32 {l}, {a}"),
    ("ds_read2_b32", "ds_read2_b32 {d2}, {l} offset0:0 offset1:64"),
]

HEADER = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define ITERS 4096
"""

KERNEL = r"""
__global__ __launch_bounds__(256) void k_%(name)s(unsigned long long* out, unsigned* sink) {
  __shared__ unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  unsigned l4 = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024;       // conflict-free dword per lane
  unsigned l8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024;
  (void)l8;
  asm volatile(
      "v_mov_b32 v8, %%2\n v_mov_b32 v9, %%3\n v_mov_b32 v10, 3\n v_mov_b32 v11, 1.5\n"
      "v_mov_b32 v12, 7\n v_mov_b32 v13, 2.5\n v_mov_b32 v14, 11\n v_mov_b32 v15, 0.75\n"
      "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n"
      "v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"
      "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
      "s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333\n"
      "s_mov_b64 vcc, 0x5555\n s_mov_b64 s[30:31], exec\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_memtime s[24:25]\n"
      "s_movk_i32 s23, %(iters)d\n"
      "s_waitcnt lgkmcnt(0)\n"
      "1:\n"
      %(body)s
      "s_sub_u32 s23, s23, 1\n"
      "s_cmp_lg_u32 s23, 0\n"
      "s_cbranch_scc1 1b\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "s_memtime s[26:27]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_sub_u32 s24, s26, s24\n s_subb_u32 s25, s27, s25\n"
      "v_mov_b32 %%0, s24\n"
      "v_add_u32 %%1, v0, v1\n v_add_u32 %%1, %%1, v2\n v_add_u32 %%1, %%1, v4\n v_add_u32 %%1, %%1, v6\n"
      : "=v"(l4), "=v"(l8)
      : "v"(l4), "v"(l8)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23",
        "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","vcc","memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = l4;
  if (l8 == 0x12345) sink[0] = l8;
}
"""


def ident(name):
    return "".join(ch if ch.isalnum() else "_" for ch in name)


def body_for(tmpl):
    if ";" in tmpl:
        return body_for_group(tmpl)
    lines = []
    for i in range(32):
        r = i % 8
        # 64-bit operands use even pairs out of v0..v7 (dest) and v10..v15 (sources)
        d2 = "v[%d:%d]" % (2 * (i % 4), 2 * (i % 4) + 1)
        ins = tmpl.format(d="v%d" % r, a="v%d" % (10 + (i % 3) * 2), b="v%d" % (12 + (i % 2) * 2), c="v14",
                          d2=d2, a2="v[10:11]", b2="v[12:13]", c2="v[14:15]", l="v8", l8="v9")
        lines.append('"%s\\n"' % ins)
        if tmpl.startswith("ds_") and i % 8 == 7:
            lines.append('"s_waitcnt lgkmcnt(0)\\n"')
    return "\n      ".join(lines)


def body_for_group(tmpl):
    """A group of instructions (';'-separated) repeated 8 times per loop trip; the table then shows the cost
    of 1/4 group per 'instruction' (32 slots per trip)."""
    parts = [t.strip() for t in tmpl.split(";")]
    lines = []
    for i in range(8):
        for t in parts:
            ins = t.format(d="v%d" % (i % 8), e="v%d" % ((i + 1) % 8), a="v%d" % (10 + (i % 3) * 2), b="v%d" % (12 + (i % 2) * 2),
                           c="v14", l="v8", l8="v9")
            lines.append('"%s\\n"' % ins)
    return "\n      ".join(lines)


def main():
    out_dir = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    build = "/tmp/valu_rates"
    os.makedirs(build, exist_ok=True)
    src = [HEADER]
    names = []
    for name, tmpl in OPS:
        if tmpl is None:
            continue
        names.append(name)
        src.append(KERNEL % {"name": ident(name), "iters": 4096, "body": body_for(tmpl)})
    src.append("typedef void (*kfn)(unsigned long long*, unsigned*);\n")
    src.append("struct Ent { const char* n; kfn f; };\nstatic Ent ents[] = {\n")
    for n in names:
        src.append('  {"%s", k_%s},\n' % (n, ident(n)))
    src.append("};\n")
    src.append(r"""
int main() {
  unsigned long long* out; unsigned* sink;
  (void)hipMalloc(&out, 8 * 4 * 256 * 8); (void)hipMalloc(&sink, 64);
  std::vector<unsigned long long> h(4 * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("ns/instr/SIMD = kernel wall time x 1024 SIMDs / wave-instructions; ticks = s_memtime per wave-instruction / waves per SIMD\n");
  printf("%-26s %9s %9s %9s | %9s %9s\n", "instruction", "ns@1w", "ns@4w", "ns@8w", "tick@4w", "tick@8w");
  for (auto& e : ents) {
    double r[3], tk[3];
    int ws[3] = {1, 4, 8};
    for (int k = 0; k < 3; k++) {
      int W = ws[k];
      int blocks = 256 * W;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, out, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      (void)hipMemcpy(h.data(), out, 8ull * blocks * 4, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> v(h.begin(), h.begin() + blocks * 4);
      std::sort(v.begin(), v.end());
      double med = (double)v[v.size() / 2];
      tk[k] = med / (4096.0 * 32.0) / W;
      r[k] = (double)ms * 1e6 * 1024.0 / ((double)blocks * 4.0 * 4096.0 * 32.0);
    }
    printf("%-26s %9.3f %9.3f %9.3f | %9.2f %9.2f\n", e.n, r[0], r[1], r[2], tk[1], tk[2]);
  }
  return 0;
}
""")
    path = os.path.join(build, "valu_rates.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    exe = os.path.join(build, "valu_rates")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", path, "-o", exe])
    if "--build-only" in sys.argv:
        return
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    os.makedirs(os.path.join(out_dir, "gpurun_out"), exist_ok=True)
    with open(os.path.join(out_dir, "gpurun_out", "valu_rates.txt"), "w") as f:
        f.write(res.stdout)


if __name__ == "__main__":
    main()
