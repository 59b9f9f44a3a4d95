"""Probe (round 6): what the decoder's token step (planes of <= 16 symbols, counts model) pays for.

Replays token steps as hand-written loops over real LDS tables on every SIMD of the chip (as tools/probes/decode_model.py:
16 symbols of count 16, the rANS state walks as in the product, ~16 lanes pop a word per token) at 8 waves per SIMD and
prints ns per token step per SIMD.  Variants: the round-5 step, the round-6 step (quarter by unsigned minimum of key - pivot,
v_cmpx + vcc word pop, scales from SGPRs, one v_cvt_pk_bf16_f32 and two stores per token pair), and the round-6 step with
parts taken out (its scalar side, its LDS reads, its stores, ...) -- what each part costs beside the others.

    python tools/probes/decode_step.py      (on the GPU box; writes gpurun_out/decode_step.txt)
"""
import os
import subprocess
import sys

HEADER = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
"""
# registers: v0 x | v1 col | v3 sl | v4 q | v5 P0 v6 PA v7 PB v8 PC (r05: v5 pm, v6-v8 raw pivots) | v9 cffff | v10 Lv
# v28-31 e4 | v15-v18 d | v19 f | v20 r | v21 t | v22 lv (token A: v32) | v23 sc | v24 out v33 | v25 voff | v26 colB
# s20 ring | s21 e | s22 i | s23 loop | s[30:31] full | s[12:13] out base | s14 scale pair | s7 perm sel | s8 trig
L1_R5 = [
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
]
L1_R6 = [
    "v_lshl_or_b32 v3, v0, 23, v9",
    "v_sub_u32_e32 v15, v3, v5",
    "v_sub_u32_e32 v16, v3, v6",
    "v_sub_u32_e32 v17, v3, v7",
    "v_sub_u32_e32 v4, v3, v8",
    "v_min3_u32 v15, v15, v16, v17",
    "v_min_u32_e32 v4, v4, v15",
    "v_and_b32_e32 v4, 0xfffff, v4",
]
READ_Q = ["ds_read_b128 v[28:31], v4", "s_waitcnt lgkmcnt(0)"]
L2 = [
    "v_sub_u32_e32 v15, v3, v28",
    "v_sub_u32_e32 v16, v3, v29",
    "v_sub_u32_e32 v17, v3, v30",
    "v_sub_u32_e32 v18, v3, v31",
    "v_min3_u32 v15, v15, v16, v17",
    "v_lshrrev_b32_e32 v0, 9, v0",
    "v_min_u32_e32 v18, v18, v15",
    "v_bfe_u32 v20, v18, 10, 10",
]
READ_LV = ["ds_read_b32 %(lv)s, v20"]
UPD = [
    "v_and_b32_e32 v19, 0x3ff, v18",
    "v_lshrrev_b32_e32 v18, 23, v18",
    "v_mad_u32_u24 v0, v0, v19, v18",
]
POP_R5 = [
    "v_cmp_lt_u32_e64 s[10:11], v0, v10",
    "s_bcnt1_i32_b64 s6, s[10:11]",
    "s_sub_i32 s21, s21, s6",
    "s_and_b32 s6, s21, 0xff",
    "s_lshl1_add_u32 s6, s6, s20",
    "s_mov_b64 exec, s[10:11]",
    "v_mbcnt_lo_u32_b32 v21, s10, 0",
    "v_mbcnt_hi_u32_b32 v21, s11, v21",
    "v_lshl_add_u32 v21, v21, 1, s6",
    "ds_read_u16 v21, v21",
    "s_waitcnt lgkmcnt(0)",
    "v_perm_b32 v0, v21, v0, s7",
    "s_mov_b64 exec, s[30:31]",
    "s_cmp_gt_i32 s21, s8",
    "s_cbranch_scc0 9f",
]
POP_R6 = [
    "v_cmpx_lt_u32_e32 vcc, v0, v10",
    "s_bcnt1_i32_b64 s6, vcc",
    "s_sub_i32 s21, s21, s6",
    "s_and_b32 s6, s21, 0xff",
    "s_lshl1_add_u32 s6, s6, s20",
    "v_mbcnt_lo_u32_b32 v21, vcc_lo, 0",
    "v_mbcnt_hi_u32_b32 v21, vcc_hi, v21",
    "v_lshl_add_u32 v21, v21, 1, s6",
    "ds_read_u16 v21, v21",
    "s_waitcnt lgkmcnt(0)",
    "v_perm_b32 v0, v21, v0, s7",
    "s_mov_b64 exec, s[30:31]",
    "s_cmp_gt_i32 s21, s8",
    "s_cbranch_scc0 9f",
]
OUT_R5 = [
    "v_readlane_b32 s9, v23, s22",
    "s_add_i32 s22, s22, 1",
    "s_and_b32 s22, s22, 63",
    "v_mul_f32_e32 v24, s9, v22",
    "v_cvt_pk_bf16_f32 v24, v24, s9",
    "global_store_short v25, v24, s[12:13]",
]
OUT_PAIR = [
    "s_lshl_b32 s9, s14, 16",
    "v_mul_f32_e32 v24, s9, v32",
    "s_and_b32 s9, s14, 0xffff0000",
    "v_mul_f32_e32 v33, s9, v22",
    "v_cvt_pk_bf16_f32 v24, v24, v33",
    "s_add_i32 s22, s22, s15",
    "global_store_short v25, v24, s[12:13]",
    "global_store_short_d16_hi v25, v24, s[12:13] offset:128",
]


def fmt(lines, **kw):
    return [ln % kw if "%(" in ln else ln for ln in lines]


def strip(lines, pred):
    return [ln for ln in lines if not pred(ln)]


def step_r5():
    return L1_R5 + READ_Q + L2[:7] + UPD[:1] + [L2[7]] + UPD[1:] + fmt(READ_LV, lv="v22") + POP_R5 + OUT_R5


def pair_r6(l1=L1_R6, pop=POP_R6, out=OUT_PAIR, read_lv=True, drop=None):
    body = []
    for lv in ("v32", "v22"):
        body += l1 + READ_Q + L2 + (fmt(READ_LV, lv=lv) if read_lv else []) + UPD + pop
    body += out
    if drop:
        body = strip(body, drop)
    return body


is_salu = lambda ln: ln.startswith("s_") and not ln.startswith("s_waitcnt") and "exec" not in ln and not ln.startswith("s_cbranch") or ln.startswith("s_cmp")
is_lds = lambda ln: ln.startswith("ds_")
is_store = lambda ln: ln.startswith("global_store")
is_wait = lambda ln: ln.startswith("s_waitcnt")

# (name, list of instructions of TWO token steps)
VARIANTS = [
    ("r05 step (27 VALU, 14 SALU)", step_r5() + step_r5()),
    ("r06 step (25.5 VALU)", pair_r6()),
    ("r06, round-5 quarter search", pair_r6(l1=L1_R5)),
    ("r06, round-5 word pop (v_cmp + 2 exec writes)", pair_r6(pop=POP_R5)),
    ("r06, round-5 output (readlane, cvt per token)", [x for lv in ("v22", "v22") for x in (L1_R6 + READ_Q + L2 + fmt(READ_LV, lv=lv) + UPD + POP_R6 + OUT_R5)]),
    ("r06 without the stores", pair_r6(drop=is_store)),
    ("r06 without the output block", pair_r6(out=[])),
    ("r06 without SALU (but the exec write)", pair_r6(drop=lambda ln: is_salu(ln) or ln.startswith("s_cbranch"))),
    ("r06 without the ring-event test + branch", pair_r6(drop=lambda ln: ln.startswith("s_cmp_gt") or ln.startswith("s_cbranch"))),
    ("r06 without the LUT read", pair_r6(read_lv=False)),
    ("r06 without LDS reads and waits (chain broken)", pair_r6(drop=lambda ln: is_lds(ln) or is_wait(ln))),
    ("r06 VALU only", pair_r6(drop=lambda ln: not ln.startswith("v_") or ln.startswith("v_cmpx"))),
    ("r06 without the word pop (mbcnt .. perm)", pair_r6(pop=["v_cmp_lt_u32_e64 s[10:11], v0, v10", "s_bcnt1_i32_b64 s6, s[10:11]", "s_sub_i32 s21, s21, s6"])),
]

KERNEL = r"""
__global__ __launch_bounds__(512) void k_%(name)s(unsigned short* gout, unsigned* sink, int r5) {
  __shared__ __attribute__((aligned(4096))) unsigned lds[3072 + 8 * 160 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* tab = lds;
  float* dq = reinterpret_cast<float*>(lds + 3072 + 8 * 160);
  if (threadIdx.x < 32) dq[threadIdx.x] = ((float)threadIdx.x - 7.0f) / 7.0f;
  typedef __attribute__((address_space(3))) unsigned* lp;
  const unsigned dqb = (unsigned)(size_t)(lp) reinterpret_cast<unsigned*>(dq);
  if (wave == 0) {
    for (int i = 0; i < 16; i++) {  // quarters of packed entries as k_decode builds them (counts model)
      const unsigned acc = 16u * i, lutad = (dqb + 4 * i) & 0x3ffu;
      tab[(i >> 2) * 256 + lane * 4 + (i & 3)] = (acc << 24) | (0x7ffffeu - (((3u - (i & 3u)) << 20) | (lutad << 10) | 32u));
    }
  }
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = lane; i < 160; i += 64) { h = h * 1664525u + 1013904223u; lds[3072 + wave * 160 + i] = h; }
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(lp)tab;
  const unsigned col = base + 16u * lane;
  const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 12288u + 640u * (unsigned)wave));
  unsigned p0 = tab[lane * 4], pA = tab[1 * 256 + lane * 4], pB = tab[2 * 256 + lane * 4], pC = tab[3 * 256 + lane * 4];
  if (!r5) {
    p0 = (p0 & 0xff800000u) | (0x7ffffeu - ((3u << 20) | col));
    pA = (pA & 0xff800000u) | (0x7ffffeu - ((2u << 20) | (col + 1024u)));
    pB = (pB & 0xff800000u) | (0x7ffffeu - ((1u << 20) | (col + 2048u)));
    pC = (pC & 0xff800000u) | (0x7ffffeu - ((0u << 20) | (col + 3072u)));
  }
  unsigned short* gbase = gout + ((size_t)blockIdx.x * 8 + wave) * 128;
  const unsigned glo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)gbase);
  const unsigned ghi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)gbase >> 32));
  unsigned xout;
  asm volatile(
      "v_mov_b32 v0, %%1\n v_mov_b32 v1, %%2\n v_mov_b32 v5, %%3\n v_mov_b32 v6, %%4\n v_mov_b32 v7, %%5\n v_mov_b32 v8, %%6\n"
      "v_mov_b32 v9, 0x7ffffe\n v_mov_b32 v10, 0x8000\n v_add_u32 v26, 0x800, v1\n v_mov_b32 v23, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v32, 1.0\n"
      "v_mov_b32 v20, %%7\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v21, 0\n"
      "v_mbcnt_lo_u32_b32 v25, -1, 0\n v_mbcnt_hi_u32_b32 v25, -1, v25\n v_lshlrev_b32 v25, 1, v25\n"
      "s_mov_b32 s20, %%8\n s_mov_b32 s21, 0x100000\n s_mov_b32 s22, 0\n s_mov_b64 s[30:31], exec\n s_mov_b32 s7, 0x01000504\n s_mov_b32 s8, 0\n"
      "s_mov_b32 s12, %%9\n s_mov_b32 s13, %%10\n s_mov_b32 s14, 0x3f803f80\n s_mov_b32 s15, 0\n s_mov_b64 s[10:11], 0\n"
      "s_movk_i32 s23, %(iters)d\n"
      "1:\n"
      %(body)s
      "s_sub_u32 s23, s23, 1\n"
      "s_cmp_lg_u32 s23, 0\n"
      "s_cbranch_scc1 1b\n"
      "9:\n"
      "s_mov_b64 exec, s[30:31]\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "v_mov_b32 %%0, v0\n"
      : "=v"(xout)
      : "v"(0x8000u + (h & 0x7fffu)), "v"(col), "v"(p0), "v"(pA), "v"(pB), "v"(pC), "v"(dqb), "s"(ring), "s"(glo), "s"(ghi)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20",
        "v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33",
        "s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s20","s21","s22","s23","s30","s31","vcc","scc","memory");
  if (xout == 0x12345) sink[0] = xout;
}
"""

PAIRS = 4   # token pairs per loop trip

L2_CMPX = [  # levels 3-4 by exec-predicated moves (the CDF16 form), then key - entry
    "v_cmpx_le_u32_e32 vcc, v30, v3",
    "v_mov_b32_e32 v28, v30",
    "v_mov_b32_e32 v29, v31",
    "s_mov_b64 exec, s[30:31]",
    "v_cmpx_le_u32_e32 vcc, v29, v3",
    "v_mov_b32_e32 v28, v29",
    "s_mov_b64 exec, s[30:31]",
    "v_lshrrev_b32_e32 v0, 9, v0",
    "v_sub_u32_e32 v18, v3, v28",
    "v_bfe_u32 v20, v18, 10, 10",
]
if "--loo" in sys.argv:
    base = pair_r6(l1=L1_R5)
    half = len(base[:-len(OUT_PAIR)]) // 2
    VARIANTS = [("r06 + round-5 quarter search", base)]
    for k in range(half):
        ln = base[k]
        if ln.startswith("s_waitcnt") or ln.startswith("s_cbranch"):
            continue
        v = list(base)
        del v[half + k]
        del v[k]
        VARIANTS.append(("  - " + ln[:50], v))
    for k in range(len(OUT_PAIR)):
        v = list(base)
        del v[2 * half + k]
        VARIANTS.append(("  - " + OUT_PAIR[k][:50], v))
elif "--alt" in sys.argv:
    def pair_alt(l1=L1_R5, l2=L2, pop=POP_R6, out=OUT_PAIR):
        body = []
        for lv in ("v32", "v22"):
            body += l1 + READ_Q + l2 + fmt(READ_LV, lv=lv) + UPD + pop
        return body + out
    VARIANTS = [
        ("r06 + round-5 quarter search", pair_alt()),
        ("... + levels 3-4 by v_cmpx / v_mov", pair_alt(l2=L2_CMPX)),
        ("r06 (min-trick quarter)", pair_alt(l1=L1_R6)),
    ]


def asm_lines(lines):
    return "\n      ".join('"%s\\n"' % ln for ln in lines)


def count(lines):
    v = sum(1 for ln in lines if ln.startswith("v_"))
    s = sum(1 for ln in lines if ln.startswith("s_") and not ln.startswith("s_waitcnt"))
    d = sum(1 for ln in lines if ln.startswith("ds_"))
    return v / 2.0, s / 2.0, d / 2.0


def main():
    out_dir = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    build = "/tmp/decode_step"
    os.makedirs(build, exist_ok=True)
    iters = 512
    src = [HEADER]
    for k, (name, pair) in enumerate(VARIANTS):
        src.append(KERNEL % {"name": "v%d" % k, "iters": iters, "body": asm_lines(pair * PAIRS)})
    ents = ", ".join('{"%s", k_v%d, %d, %.1f, %.1f, %.1f}' % ((name, k, 1 if k == 0 else 0) + count(pair)) for k, (name, pair) in enumerate(VARIANTS))
    src.append(r"""
typedef void (*kfn)(unsigned short*, unsigned*, int);
int main(int argc, char** argv) {
  unsigned short* out; unsigned* sink;
  (void)hipMalloc(&out, 2ull * 256 * 4 * 8 * 128 * 2); (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct { const char* n; kfn f; int r5; double v, s, d; } ents[] = {@ENTS@};
  printf("ns per token step per SIMD = kernel wall time x 1024 SIMDs / (waves x tokens per wave)\n");
  printf("%-58s %5s %5s %4s %10s %10s %10s\n", "token step", "VALU", "SALU", "LDS", "ns@2w/SIMD", "ns@4w", "ns@8w");
  for (int rounds = 0; rounds < 2; rounds++)
  for (auto& e : ents) {
    double r[3];
    int ws[3] = {1, 2, 4};
    for (int k = 0; k < 3; k++) {
      int blocks = 256 * ws[k];
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(512), 0, 0, out, sink, e.r5);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      r[k] = (double)best * 1e6 * 1024.0 / ((double)blocks * 8.0 * @ITERS@.0 * 2.0 * @PAIRS@.0);
    }
    printf("%-58s %5.1f %5.1f %4.1f %10.2f %10.2f %10.2f\n", e.n, e.v, e.s, e.d, r[0], r[1], r[2]);
  }
  printf("status %d\n", (int)hipDeviceSynchronize());
  return 0;
}
""".replace("@ITERS@", str(iters)).replace("@PAIRS@", str(PAIRS)).replace("@ENTS@", ents))
    path = os.path.join(build, "decode_step.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    exe = os.path.join(build, "decode_step")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", path, "-o", exe])
    if "--build-only" in sys.argv:
        return
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    os.makedirs(os.path.join(out_dir, "gpurun_out"), exist_ok=True)
    with open(os.path.join(out_dir, "gpurun_out", "decode_step.txt"), "w") as f:
        f.write(res.stdout)


if __name__ == "__main__":
    main()
