"""Probe: would a direct slot -> symbol LUT beat the decoder's search?  

k_decode's token step on planes of <= 16 symbols is 27 VALU: 20 of them find the symbol (two levels on register pivots,
one ds_read_b128, an unsigned-min over the quarter).  A LUT indexed by the slot would make that 11 -- but 256 slots x 4 bit
= 128 B per channel = 8 KiB per wave (+ a 4 KiB entry table), which leaves 3 waves per SIMD where the search runs 8.  The
decoder's token is a dependent chain (state -> slot -> table -> state), so fewer waves means less of that latency
hidden.  This probe replays BOTH token steps as hand-written loops over real LDS tables (16 symbols of count 16: the state
walks exactly as in the product, ~16 lanes pop a word per token) and times them at 2, 4 and 8 waves per SIMD on every
SIMD of the chip; the LUT form can only ever run at 2 - 3 (4 with 16-bit entries and a smaller ring).

    python tools/probes/decode_model.py      (on the GPU box; writes gpurun_out/decode_model.txt)
"""
import os
import subprocess
import sys

HEADER = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
"""

# registers: v0 x | v1 col (search: lane*16 + table base; lut: lane*4 + lut base) | v2 tabcol (lut: entry column) | v3 sl/slot
# v4 q | v5 pm | v6 pA v7 pB v8 pC | v9 cffff | v10 Lv | v11-v14 e4 | v15-v18 d | v19 f | v20 r | v21 t | v22 lv | v23 sc
# v24 out | v25 voff | v26 colB | v27 tmp
# s20 ring | s21 e | s22 i | s23 loop | s[30:31] full | s[12:15] rsrc? (store through global_store_short to a per-wave row)
STEP_SEARCH = [
    "v_lshl_or_b32 v3, v0, 23, v9",
    "v_mov_b32_e32 v4, v1",
    "v_mov_b32_e32 v5, v6",
    "v_cmpx_le_u32_e32 vcc, v7, v3",
    "v_mov_b32_e32 v4, v26",
    "v_mov_b32_e32 v5, v8",
    "s_mov_b64 exec, s[30:31]",
    "v_cmpx_le_u32_e32 vcc, v5, v3",
    "v_add_u32_e32 v4, 0x400, v4",
    "s_mov_b64 exec, s[30:31]",
    "ds_read_b128 v[28:31], v4",
    "s_waitcnt lgkmcnt(0)",
    "v_sub_u32_e32 v15, v3, v28",
    "v_sub_u32_e32 v16, v3, v29",
    "v_sub_u32_e32 v17, v3, v30",
    "v_sub_u32_e32 v18, v3, v31",
    "v_min3_u32 v15, v15, v16, v17",
    "v_lshrrev_b32_e32 v0, 9, v0",
    "v_min_u32_e32 v18, v18, v15",
    "v_and_b32_e32 v19, 0x3ff, v18",
    "v_bfe_u32 v20, v18, 10, 10",
    "v_lshrrev_b32_e32 v18, 23, v18",
    "v_mad_u32_u24 v0, v0, v19, v18",
    "v_cmp_lt_u32_e64 s[10:11], v0, v10",
]
STEP_LUT = [
    "v_and_b32_e32 v3, 0x1ff, v0",              # slot
    "v_lshrrev_b32_e32 v4, 4, v3",              # dword of the LUT: units 8 k .. 8 k + 7
    "v_lshl_add_u32 v4, v4, 8, v1",
    "ds_read_b32 v11, v4",
    "v_and_b32_e32 v5, 14, v3",                 # nibble position: ((slot >> 1) & 7) * 4
    "v_lshlrev_b32_e32 v5, 1, v5",
    "s_waitcnt lgkmcnt(0)",
    "v_bfe_u32 v12, v11, v5, 4",                # symbol
    "v_lshl_add_u32 v13, v12, 8, v2",           # its entry: start << 16 | freq
    "ds_read_b32 v14, v13",
    "v_lshl_add_u32 v20, v12, 2, v27",          # its dequantisation LUT entry
    "v_lshrrev_b32_e32 v0, 9, v0",
    "s_waitcnt lgkmcnt(0)",
    "v_sub_u32_sdwa v18, v3, v14 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
    "v_and_b32_e32 v19, 0xffff, v14",
    "v_mad_u32_u24 v0, v0, v19, v18",
    "v_cmp_lt_u32_e64 s[10:11], v0, v10",
]
POP_AND_OUT = [
    "s_bcnt1_i32_b64 s6, s[10:11]",
    "s_sub_i32 s21, s21, s6",
    "ds_read_b32 v22, v20",
    "s_lshl_b32 s6, s21, 1",
    "s_and_b32 s6, s6, 0x1fe",
    "s_add_i32 s6, s6, s20",
    "s_mov_b64 exec, s[10:11]",
    "v_mbcnt_lo_u32_b32 v21, s10, 0",
    "v_mbcnt_hi_u32_b32 v21, s11, v21",
    "v_lshl_add_u32 v21, v21, 1, s6",
    "ds_read_u16 v21, v21",
    "s_waitcnt lgkmcnt(0)",
    "v_perm_b32 v0, v21, v0, s7",
    "s_mov_b64 exec, s[30:31]",
    "s_cmp_gt_i32 s21, s8",                     # the ring's refill test (never taken here)
    "s_cbranch_scc0 9f",
    "v_readlane_b32 s9, v23, s22",
    "s_add_i32 s22, s22, 1",
    "s_and_b32 s22, s22, 63",
    "v_mul_f32_e32 v24, s9, v22",
    "v_cvt_pk_bf16_f32 v24, v24, s9",
    "global_store_short v25, v24, s[12:13]",
]

KERNEL = r"""
__global__ __launch_bounds__(512) void k_%(name)s(unsigned short* gout, unsigned* sink, int lut) {
  // ONE copy of the tables for the workgroup (the probe sets the occupancy by its grid, not by its LDS footprint: every
  // wave reads the same 12 KiB, each lane its own column, as in the product), a 640-byte word ring per wave, the
  // dequantisation LUT behind them
  __shared__ __attribute__((aligned(4096))) unsigned lds[3072 + 8 * 160 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* tab = lds;
  float* dq = reinterpret_cast<float*>(lds + 3072 + 8 * 160);
  if (threadIdx.x < 32) dq[threadIdx.x] = ((float)threadIdx.x - 7.0f) / 7.0f;
  if (wave != 0) {
  } else if (lut) {
    for (int k = 0; k < 32; k++) tab[k * 64 + lane] = 0x11111111u * (unsigned)(k >> 1);    // LUT: unit -> symbol (16 units each)
    for (int s = 0; s < 16; s++) tab[2048 + s * 64 + lane] = ((32u * s) << 16) | 32u;       // entries: start << 16 | freq
  } else {
    for (int i = 0; i < 16; i++) {  // quarters of packed entries as k_decode builds them (counts model)
      const unsigned acc = 16u * i, lutad = (unsigned)(4 * i) & 0x3ffu;  // (the product keeps the LUTs in LDS' first KiB; here the field points into the table: timing only)
      tab[(i >> 2) * 256 + lane * 4 + (i & 3)] = (acc << 24) | (0x7ffffeu - (((3u - (i & 3u)) << 20) | (lutad << 10) | 32u));
    }
  }
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = lane; i < 160; i += 64) { h = h * 1664525u + 1013904223u; lds[3072 + wave * 160 + i] = h; }
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned* lp;
  const unsigned base = (unsigned)(size_t)(lp)tab;
  const unsigned col = lut ? base + 4u * lane : base + 16u * lane;
  const unsigned tabcol = base + 8192u + 4u * lane;
  const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 12288u + 640u * (unsigned)wave));
  const unsigned dqb = (unsigned)(size_t)(lp) reinterpret_cast<unsigned*>(dq);
  unsigned pA = tab[1 * 256 + lane * 4], pB = tab[2 * 256 + lane * 4], pC = tab[3 * 256 + lane * 4];
  unsigned short* gbase = gout + ((size_t)blockIdx.x * 8 + wave) * 64;
  const unsigned glo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)gbase);
  const unsigned ghi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)gbase >> 32));
  unsigned xout;
  asm volatile(
      "v_mov_b32 v0, %%1\n v_mov_b32 v1, %%2\n v_mov_b32 v2, %%3\n v_mov_b32 v6, %%4\n v_mov_b32 v7, %%5\n v_mov_b32 v8, %%6\n"
      "v_mov_b32 v9, 0x7ffffe\n v_mov_b32 v10, 0x8000\n v_add_u32 v26, 0x800, v1\n v_mov_b32 v27, %%7\n v_mov_b32 v23, 1.0\n"
      "v_mbcnt_lo_u32_b32 v25, -1, 0\n v_mbcnt_hi_u32_b32 v25, -1, v25\n v_lshlrev_b32 v25, 1, v25\n"
      "s_mov_b32 s20, %%8\n s_mov_b32 s21, 0x100000\n s_mov_b32 s22, 0\n s_mov_b64 s[30:31], exec\n s_mov_b32 s7, 0x01000504\n s_mov_b32 s8, 0\n"
      "s_mov_b32 s12, %%9\n s_mov_b32 s13, %%10\n"
      "s_movk_i32 s23, %(iters)d\n"
      "1:\n"
      %(body)s
      "s_sub_u32 s23, s23, 1\n"
      "s_cmp_lg_u32 s23, 0\n"
      "s_cbranch_scc1 1b\n"
      "9:\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "v_mov_b32 %%0, v0\n"
      : "=v"(xout)
      : "v"(0x8000u + (h & 0x7fffu)), "v"(col), "v"(tabcol), "v"(pA), "v"(pB), "v"(pC), "v"(dqb), "s"(ring), "s"(glo), "s"(ghi)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20",
        "v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31",
        "s6","s7","s8","s9","s10","s11","s12","s13","s20","s21","s22","s23","s30","s31","vcc","scc","memory");
  if (xout == 0x12345) sink[0] = xout;
}
"""

UNROLL = 8


def asm_lines(lines):
    return "\n      ".join('"%s\\n"' % ln for ln in lines)


def main():
    out_dir = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    build = "/tmp/decode_model"
    os.makedirs(build, exist_ok=True)
    iters = 512
    src = [HEADER]
    for name, step in (("search", STEP_SEARCH), ("lut", STEP_LUT)):
        body = []
        for _ in range(UNROLL):
            body += step + POP_AND_OUT
        src.append(KERNEL % {"name": name, "iters": iters, "body": asm_lines(body)})
    src.append(r"""
typedef void (*kfn)(unsigned short*, unsigned*, int);
int main() {
  unsigned short* out; unsigned* sink;
  (void)hipMalloc(&out, 2ull * 256 * 4 * 8 * 64 * 2); (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct { const char* n; kfn f; int lut; } ents[] = {{"search (k_decode's step: 27 VALU)", k_search, 0}, {"LUT (20 VALU)", k_lut, 1}};
  printf("ns per token step per SIMD = kernel wall time x 1024 SIMDs / (waves x tokens per wave); k_decode runs 8 waves per SIMD at 4.9 KiB of LDS per wave, the LUT form needs 12.6 KiB: 3\n");
  printf("%-36s %12s %12s %12s\n", "token step", "ns@2w/SIMD", "ns@4w", "ns@8w");
  for (auto& e : ents) {
    double r[3];
    int ws[3] = {1, 2, 4};
    for (int k = 0; k < 3; k++) {
      int blocks = 256 * ws[k];
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(512), 0, 0, out, sink, e.lut);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      r[k] = (double)best * 1e6 * 1024.0 / ((double)blocks * 8.0 * @ITERS@.0 * @UNROLL@.0);
    }
    printf("%-36s %12.2f %12.2f %12.2f\n", e.n, r[0], r[1], r[2]);
  }
  printf("status %d\n", (int)hipDeviceSynchronize());
  return 0;
}
""".replace("@ITERS@", str(iters)).replace("@UNROLL@", str(UNROLL)))
    path = os.path.join(build, "decode_model.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    exe = os.path.join(build, "decode_model")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", path, "-o", exe])
    if "--build-only" in sys.argv:
        return
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    os.makedirs(os.path.join(out_dir, "gpurun_out"), exist_ok=True)
    with open(os.path.join(out_dir, "gpurun_out", "decode_model.txt"), "w") as f:
        f.write(res.stdout)


if __name__ == "__main__":
    main()
