"""Probe: what does one TOKEN STEP of the counts coder cost in issue slots -- vector by class AND scalar?

Round 4's roofline priced every VALU instruction at 4 SIMD cycles and did not price the scalar side at all
("What's weak" #2).  This probe replays the pass-2 token step of k_encode_fused (the instruction
sequence of the shipped ISA, tools/isa_cycles.py --between v_cmpx_ge_u32_sdwa, with real LDS tables, a real ring
and a real flush) as a hand-written loop, 8 waves per SIMD on every SIMD of the chip, and then takes it apart:

  pure streams   ns per wave-instruction per SIMD of s_add / s_and / s_lshl / s_bcnt1 / s_mov exec / s_cmp+s_cbranch
                 (taken and not taken) / s_nop / s_waitcnt, alone and beside a VALU stream of another wave
  step replicas  the token step as shipped; without its scalar bookkeeping; without the emit block; vector part
                 only; scalar part only; with a LEAN scalar side (running ring address, one compare);
                 and the format-v7 candidate: ONE renormalisation per token PAIR (frequencies out of 2^8,
                 state in [2^16, 2^32), quotient by mul_hi + add + shift)

Output: ns per token step per SIMD (= kernel time x 1024 SIMDs / (waves x token steps per wave)), at 4 and 8
waves per SIMD.  python tools/probes/issue_model.py  (on the GPU box; writes gpurun_out/issue_model.txt)
"""
import os
import subprocess
import sys

# ---- register plan (explicit: the probe is one asm block) -------------------------------------------------------
# v0  col      LDS byte address of table[0][lane] (dword entries, rows of 256 B)
# v1  x        rANS state
# v2-v6        entry pipeline E[0..4]  (rotating: current = E[i % 5], loaded this step = E[(i + 4) % 5])
# v32-v41      reciprocal pipeline R[0..4] (pairs; current = R[i % 5], loaded this step = R[(i + 2) % 5])
# v17-v20      symbol dwords w[0..3] (pseudo-random nibbles)
# v21 t  v22 ra  v23 ad  v24 q  v25 lane  v26 lane*4  v27 tmp  v28 F  v29 tmp2
# s20 ring base  s21 wcur  s22 flushed  s23 loop counter  s[24:27] time  s[30:31] full exec  s[12:13] global base
# s6 s7 temps  s34 wb (running ring address, lean variants)  s35 limit
E = ["v%d" % (2 + k) for k in range(5)]
R = [("v%d" % (32 + 2 * k), "v%d" % (33 + 2 * k)) for k in range(5)]
W = ["v17", "v18", "v19", "v20"]
RTAB_OFF = 8 * 4736  # byte offset of the reciprocal table behind 8 waves' tables + rings

UNROLL = 40


def flush_block(label):
    """the ring's flush as shipped (k_encode_counts.h flush_ring): every 128 words"""
    return [
        "s_sub_i32 s6, s21, s22",
        "s_cmpk_lt_u32 s6, 0x80",
        "s_cbranch_scc1 %s" % label,
        "s_bitcmp1_b32 s22, 7",
        "s_cselect_b64 s[36:37], -1, 0",
        "s_addk_i32 s6, 0xff80",
        "v_cmp_gt_u32_e32 vcc, s6, v25",
        "s_and_b64 s[36:37], s[36:37], vcc",
        "s_and_saveexec_b64 s[6:7], s[36:37]",
        "v_lshl_add_u32 v27, v25, 1, s20",
        "ds_read_u16 v29, v27 offset:512",
        "s_waitcnt lgkmcnt(0)",
        "ds_write_b16 v27, v29",
        "s_or_b64 exec, exec, s[6:7]",
        "s_lshl_b32 s6, s22, 1",
        "s_and_b32 s7, s6, 0x1fc",
        "v_add_u32_e32 v27, s7, v26",
        "v_add_u32_e32 v27, s20, v27",
        "ds_read_b32 v27, v27",
        "s_addk_i32 s22, 0x80",
        "s_waitcnt lgkmcnt(0)",
        "global_store_dword v26, v27, s[12:13]",
        "%s:" % label,
    ]


def sym_addr(i, dst="v23"):
    """row address of the entry four tokens on: nibble (i % 8) of w[(i // 8) % 4]"""
    pos = 4 * (i % 8)
    w = W[(i // 8) % 4]
    if pos == 8:
        return ["v_and_or_b32 %s, %s, s33, v0" % (dst, w)]  # s33 = 0xf00
    if pos == 28:
        return ["v_lshrrev_b32_e32 %s, 28, %s" % (dst, w), "v_lshl_add_u32 %s, %s, 8, v0" % (dst, dst)]
    return ["v_bfe_u32 %s, %s, %d, 4" % (dst, w, pos), "v_lshl_add_u32 %s, %s, 8, v0" % (dst, dst)]


def step_v6(i, salu="full", emit=True, valu=True, lds=True, nop=True, flush=True):
    """one token step of the shipped coder (nibble plane).  i = position in the unrolled body."""
    e0, e2, e4 = E[i % 5], E[(i + 2) % 5], E[(i + 4) % 5]
    r0, r2 = R[i % 5], R[(i + 2) % 5]
    o = []
    if valu:
        o.append("v_lshrrev_b32_e32 v22, 20, %s" % e2)
    if emit:
        if valu:
            o.append("v_cmpx_ge_u32_sdwa vcc, v1, %s src0_sel:WORD_1 src1_sel:WORD_1" % e0)
        if salu != "none":
            o.append("s_bcnt1_i32_b64 s7, vcc")
            if nop:
                o.append("s_nop 0")
        if valu:
            o += ["v_mbcnt_lo_u32_b32 v21, vcc_lo, 0", "v_mbcnt_hi_u32_b32 v21, vcc_hi, v21"]
            o.append("v_lshl_add_u32 v21, v21, 1, %s" % ("s34" if salu in ("lean", "lean2") else "s6"))
            if lds:
                o.append("ds_write_b16 v21, v1")
            o.append("v_lshrrev_b32_e32 v1, 16, v1")
        if valu or salu != "none":
            o.append("s_mov_b64 exec, s[30:31]")
    if valu:
        o += sym_addr(i)
    if lds and valu:
        o.append("ds_read_b64 v[%s:%s], v22 offset:%d" % (r2[0][1:], r2[1][1:], RTAB_OFF))
        o.append("ds_read_b32 %s, v23" % e4)
    if salu == "full":
        o.append("s_add_i32 s21, s7, s21")
        if flush:
            o += flush_block("L%d_%%=" % i)
        o.append("s_lshl_b32 s6, s21, 1")
    elif salu == "lean":
        # running ring address: wb += 2 * cnt; one compare against the flush limit; the flush re-bases
        o.append("s_lshl1_add_u32 s34, s7, s34")
        if flush:
            o += lean_flush("L%d_%%=" % i)
    elif salu == "lean2":
        o.append("s_lshl1_add_u32 s34, s7, s34")
        if i % 2 == 1:
            o += lean2_check(i)
    if valu:
        o += [
            "v_mul_hi_u32 v24, v1, %s" % r0[0],
            "v_lshrrev_b32_sdwa v24, %s, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" % r0[1],
            "v_mad_u32_u24 v1, v24, %s, v1" % r0[1],
            "v_add_u32_sdwa v1, v1, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" % e0,
        ]
    if salu == "full":
        o += ["s_and_b32 s6, s6, 0x1fe", "s_add_i32 s6, s6, s20"]
    if lds and valu:
        o.append("s_waitcnt lgkmcnt(2)")
    return o


TAIL = []  # out-of-line blocks of the body being generated (behind the loop)

# ---- round 6: a wave's TWO streams interleaved in one loop ------------------------------------------
# Stream B = the shipped step (lean2) on a second register set; the emit blocks (exec = emitting lanes) and the flush
# tests stay whole, everything between them alternates A / B instruction by instruction, so that the dependent chains of
# one stream (v_mul_hi -> shift -> mad -> add of the state; address -> load of the entry) issue in the other's shadow.
# Both streams read the wave's table and stage into the wave's ring region (timing only: the data are synthetic).
B_MAP = {"v1": "v7", "v2": "v8", "v3": "v9", "v4": "v10", "v5": "v11", "v6": "v12",
         "v17": "v13", "v18": "v14", "v19": "v15", "v20": "v16", "v21": "v52", "v22": "v53", "v23": "v54", "v24": "v55",
         "s34": "s54", "s7": "s57", "s22": "s56"}
for _k in range(5):
    B_MAP["v%d" % (32 + 2 * _k)] = "v%d" % (42 + 2 * _k)
    B_MAP["v%d" % (33 + 2 * _k)] = "v%d" % (43 + 2 * _k)


def to_b(line):
    import re

    def sub(m):
        tok = m.group(0)
        return B_MAP.get(tok, tok)
    line = re.sub(r"\bv\[(\d+):(\d+)\]", lambda m: "v[%s:%s]" % (B_MAP["v" + m.group(1)][1:], B_MAP["v" + m.group(2)][1:])
                  if "v" + m.group(1) in B_MAP else m.group(0), line)
    return re.sub(r"\b[vs]\d+\b", sub, line)


def step_sections(i, tag):
    """the shipped step (lean2, no s_nop) of token i as sections: pre | emit (atomic) | addr + loads | flush (atomic) | put"""
    e0, e2, e4 = E[i % 5], E[(i + 2) % 5], E[(i + 4) % 5]
    r0, r2 = R[i % 5], R[(i + 2) % 5]
    pre = ["v_lshrrev_b32_e32 v22, 20, %s" % e2]
    emit = ["v_cmpx_ge_u32_sdwa vcc, v1, %s src0_sel:WORD_1 src1_sel:WORD_1" % e0,
            "s_bcnt1_i32_b64 s7, vcc",
            "v_mbcnt_lo_u32_b32 v21, vcc_lo, 0", "v_mbcnt_hi_u32_b32 v21, vcc_hi, v21",
            "v_lshl_add_u32 v21, v21, 1, s34",
            "ds_write_b16 v21, v1",
            "v_lshrrev_b32_e32 v1, 16, v1",
            "s_mov_b64 exec, s[30:31]"]
    loads = sym_addr(i) + ["ds_read_b64 v[%s:%s], v22 offset:%d" % (r2[0][1:], r2[1][1:], RTAB_OFF),
                           "ds_read_b32 %s, v23" % e4]
    flush = ["s_lshl1_add_u32 s34, s7, s34"]
    if i % 2 == 1:
        fl, back = "F%s%d_%%=" % (tag, i), "B%s%d_%%=" % (tag, i)
        blk = ["%s:" % fl, "v_add_u32_e32 v27, s20, v26", "ds_read_b32 v29, v27", "ds_read_b32 v28, v27 offset:256",
               "s_sub_u32 s6, s34, s35", "v_cmp_gt_u32_e32 vcc, s6, v26", "s_waitcnt lgkmcnt(0)",
               "global_store_dword v26, v29, s[12:13]", "s_and_saveexec_b64 s[6:7], vcc", "ds_write_b32 v27, v28",
               "s_or_b64 exec, exec, s[6:7]", "s_addk_i32 s34, 0xff00", "s_addk_i32 s22, 0x80", "s_branch %s" % back]
        TAIL.extend(blk if tag == "A" else [to_b(x) if not x.endswith(":") and not x.startswith("s_branch") else x for x in blk])
        flush += ["s_cmp_lt_u32 s34, s35", "s_cbranch_scc0 %s" % fl, "%s:" % back]
    put = ["v_mul_hi_u32 v24, v1, %s" % r0[0],
           "v_lshrrev_b32_sdwa v24, %s, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" % r0[1],
           "v_mad_u32_u24 v1, v24, %s, v1" % r0[1],
           "v_add_u32_sdwa v1, v1, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" % e0]
    secs = [pre, emit, loads, flush, put]
    if tag == "B":
        secs = [[(x if (x.endswith(":") or x.startswith("s_cbranch")) else to_b(x)) for x in sec] for sec in secs]
        # (the flush test's compare reads B's cursor; its label lines and branch targets stay)
        secs[3] = [to_b(x) if x.startswith("s_cmp") or x.startswith("s_lshl1") else x for x in secs[3]]
    return secs


def zipped(a, b):
    out = []
    for k in range(max(len(a), len(b))):
        if k < len(a):
            out.append(a[k])
        if k < len(b):
            out.append(b[k])
    return out


def step_two_streams(i, interleave=True):
    A, Bs = step_sections(i, "A"), step_sections(i, "B")
    if not interleave:  # the two streams one after the other inside the step (the control: same work, no interleaving)
        return sum(A, []) + ["s_waitcnt lgkmcnt(2)"] + sum(Bs, []) + ["s_waitcnt lgkmcnt(2)"]
    o = zipped(A[0], Bs[0]) + A[1] + Bs[1] + zipped(A[2], Bs[2]) + A[3] + Bs[3] + zipped(A[4], Bs[4])
    o.append("s_waitcnt lgkmcnt(4)")
    return o


def lean2_check(i):
    """round-5 shipped form: linear 256-word buffer, the test every SECOND token, the flush out of line (the common
    path falls through an untaken branch)"""
    fl, back = "F%d_%%=" % i, "B%d_%%=" % i
    TAIL.extend([
        "%s:" % fl,
        "v_add_u32_e32 v27, s20, v26",
        "ds_read_b32 v29, v27",
        "ds_read_b32 v28, v27 offset:256",
        "s_sub_u32 s6, s34, s35",
        "v_cmp_gt_u32_e32 vcc, s6, v26",
        "s_waitcnt lgkmcnt(0)",
        "global_store_dword v26, v29, s[12:13]",
        "s_and_saveexec_b64 s[6:7], vcc",
        "ds_write_b32 v27, v28",
        "s_or_b64 exec, exec, s[6:7]",
        "s_addk_i32 s34, 0xff00",
        "s_addk_i32 s22, 0x80",
        "s_branch %s" % back,
    ])
    return ["s_cmp_lt_u32 s34, s35", "s_cbranch_scc0 %s" % fl, "%s:" % back]


def lean_flush(label):
    """linear staging buffer of 128 + 64 words: flush words [0, 128) when the cursor passes them, move the < 64
    words behind them down, re-base the cursor"""
    return [
        "s_cmp_lt_u32 s34, s35",  # s35 = ring + 256
        "s_cbranch_scc1 %s" % label,
        "v_add_u32_e32 v27, s20, v26",
        "ds_read_b32 v29, v27",
        "s_sub_u32 s6, s34, s35",  # bytes past the block
        "s_lshr_b32 s6, s6, 1",
        "v_cmp_gt_u32_e32 vcc, s6, v25",
        "s_and_saveexec_b64 s[6:7], vcc",
        "v_lshl_add_u32 v27, v25, 1, s20",
        "ds_read_u16 v28, v27 offset:256",
        "s_waitcnt lgkmcnt(0)",
        "ds_write_b16 v27, v28",
        "s_or_b64 exec, exec, s[6:7]",
        "s_addk_i32 s34, 0xff00",
        "s_addk_i32 s22, 0x80",
        "s_waitcnt lgkmcnt(0)",
        "global_store_dword v26, v29, s[12:13]",
        "%s:" % label,
    ]


def pair_v7(i, salu="full"):
    """format-v7 candidate: tokens i and i + 1 of the body share ONE renormalisation.  Entry = count << 24 | start
    (frequencies out of 2^8), reciprocal entry = {m_lo, (256 - f) | l << 24}: q = (mulhi(x, m_lo) + x) >> l."""
    o = []
    ea, eb = E[i % 5], E[(i + 1) % 5]
    ra_, rb_ = R[i % 5], R[(i + 1) % 5]
    # reciprocal addresses of the entries two tokens on (both tokens of the NEXT pair)
    o.append("v_lshrrev_b32_e32 v22, 21, %s" % E[(i + 2) % 5])
    o.append("v_lshrrev_b32_e32 v30, 21, %s" % E[(i + 3) % 5])
    o.append("v_mul_u32_u24_sdwa v28, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" % (ea, eb))
    o.append("v_cmpx_ge_u32_sdwa vcc, v1, v28 src0_sel:WORD_1 src1_sel:DWORD")
    o.append("s_bcnt1_i32_b64 s7, vcc")
    if salu == "full":
        o.append("s_nop 0")
    o += ["v_mbcnt_lo_u32_b32 v21, vcc_lo, 0", "v_mbcnt_hi_u32_b32 v21, vcc_hi, v21"]
    o.append("v_lshl_add_u32 v21, v21, 1, %s" % ("s34" if salu == "lean" else "s6"))
    o.append("ds_write_b16 v21, v1")
    o.append("v_lshrrev_b32_e32 v1, 16, v1")
    o.append("s_mov_b64 exec, s[30:31]")
    o += sym_addr(i, "v23") + sym_addr(i + 1, "v31")
    r2, r3 = R[(i + 2) % 5], R[(i + 3) % 5]
    o.append("ds_read_b64 v[%s:%s], v22 offset:%d" % (r2[0][1:], r2[1][1:], RTAB_OFF))
    o.append("ds_read_b64 v[%s:%s], v30 offset:%d" % (r3[0][1:], r3[1][1:], RTAB_OFF))
    # (the entry pipeline of a pair body is 5 deep as well: the entries four and five tokens on)
    o.append("ds_read_b32 %s, v23" % E[(i + 4) % 5])
    if salu == "full":
        o.append("s_add_i32 s21, s7, s21")
        o += flush_block("L%d_%%=" % i)
        o.append("s_lshl_b32 s6, s21, 1")
    else:
        o.append("s_lshl1_add_u32 s34, s7, s34")
        o += lean_flush("L%d_%%=" % i)
    for (e, r) in ((ea, ra_), (eb, rb_)):
        o += [
            "v_mul_hi_u32 v24, v1, %s" % r[0],
            "v_add_u32_e32 v24, v24, v1",
            "v_lshrrev_b32_sdwa v24, %s, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" % r[1],
            "v_mad_u32_u24 v1, v24, %s, v1" % r[1],
            "v_add_u32_sdwa v1, v1, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" % e,
        ]
        if e is ea:
            # the second entry of the pair after next: its slot in the rotation is free once `ea` has been used
            o.append("ds_read_b32 %s, v31" % E[(i + 5) % 5])
    if salu == "full":
        o += ["s_and_b32 s6, s6, 0x1fe", "s_add_i32 s6, s6, s20"]
    o.append("s_waitcnt lgkmcnt(2)")
    return o


def body_steps(kind):
    o = []
    if kind.startswith("pair"):
        for i in range(0, UNROLL, 2):
            o += pair_v7(i, "lean" if kind.endswith("lean") else "full")
        return o
    for i in range(UNROLL):
        if kind == "v6":
            o += step_v6(i)
        elif kind == "v6_nonop":
            o += step_v6(i, nop=False)
        elif kind == "v6_lean":
            o += step_v6(i, salu="lean", nop=False)
        elif kind == "v6_lean2":
            o += step_v6(i, salu="lean2", nop=False)
        elif kind == "v6_nosalu":
            o += step_v6(i, salu="none")
        elif kind == "v6_noflush":
            o += step_v6(i, flush=False)
        elif kind == "v6_noemit":
            o += step_v6(i, emit=False, salu="none")
        elif kind == "v6_valu_only":
            o += step_v6(i, salu="none", lds=False)
        elif kind == "v6_salu_only":
            o += step_v6(i, valu=False, lds=False, flush=False)
        elif kind == "two_streams":
            o += step_two_streams(i)
        elif kind == "two_streams_serial":
            o += step_two_streams(i, interleave=False)
        else:
            raise ValueError(kind)
    return o


# pure streams: 32 instructions per loop trip (UNROLL does not apply: tokens per trip = 32)
PURE = [
    ("s_add_u32", ["s_add_u32 s%d, s%d, 1" % (40 + k % 8, 40 + k % 8) for k in range(32)]),
    ("s_and_b32", ["s_and_b32 s%d, s%d, 0x1fe" % (40 + k % 8, 40 + (k + 1) % 8) for k in range(32)]),
    ("s_lshl1_add_u32", ["s_lshl1_add_u32 s%d, s%d, s%d" % (40 + k % 8, 40 + (k + 3) % 8, 40 + k % 8) for k in range(32)]),
    ("s_bcnt1_i32_b64", ["s_bcnt1_i32_b64 s%d, vcc" % (40 + k % 8) for k in range(32)]),
    ("s_mov_b64 exec", ["s_mov_b64 exec, s[30:31]" for k in range(32)]),
    ("s_nop 0", ["s_nop 0" for k in range(32)]),
    ("s_waitcnt lgkmcnt(0) (idle)", ["s_waitcnt lgkmcnt(0)" for k in range(32)]),
    ("s_cmp+cbranch not taken (x16)", sum([["s_cmp_eq_u32 s40, s41", "s_cbranch_scc1 9f"] for k in range(16)], []) + ["9:"]),
    ("s_cmp+cbranch taken (x16)", sum([["s_cmp_lg_u32 s40, s41", "s_cbranch_scc1 %df" % (100 + k), "s_nop 0", "%d:" % (100 + k)] for k in range(16)], [])),
    ("v_mad_u32_u24 (x32)", ["v_mad_u32_u24 v%d, v17, v18, v19" % (48 + k % 8) for k in range(32)]),
    ("v_add_u32 (x32)", ["v_add_u32_e32 v%d, v17, v18" % (48 + k % 8) for k in range(32)]),
    ("16 v_mad + 16 s_add", sum([["v_mad_u32_u24 v%d, v17, v18, v19" % (48 + k % 8), "s_add_u32 s%d, s%d, 1" % (40 + k % 8, 40 + k % 8)] for k in range(16)], [])),
    ("16 v_mad + 32 s_add", sum([["v_mad_u32_u24 v%d, v17, v18, v19" % (48 + k % 8), "s_add_u32 s%d, s%d, 1" % (40 + k % 8, 40 + k % 8), "s_add_u32 s%d, s%d, 1" % (40 + (k + 4) % 8, 40 + (k + 4) % 8)] for k in range(16)], [])),
    ("16 v_mad + 48 s_add", sum([["v_mad_u32_u24 v%d, v17, v18, v19" % (48 + k % 8)] + ["s_add_u32 s%d, s%d, 1" % (40 + (k + j) % 8, 40 + (k + j) % 8) for j in range(3)] for k in range(16)], [])),
    ("16 v_add + 16 s_add", sum([["v_add_u32_e32 v%d, v17, v18" % (48 + k % 8), "s_add_u32 s%d, s%d, 1" % (40 + k % 8, 40 + k % 8)] for k in range(16)], [])),
    ("16 v_mad + 16 s_mov exec", sum([["v_mad_u32_u24 v%d, v17, v18, v19" % (48 + k % 8), "s_mov_b64 exec, s[30:31]"] for k in range(16)], [])),
    ("16 (v_cmpx + s_mov exec)", sum([["v_cmpx_ge_u32_e32 vcc, v17, v18", "s_mov_b64 exec, s[30:31]"] for k in range(16)], [])),
    ("16 (v_cmpx + v_add + s_mov exec)", sum([["v_cmpx_ge_u32_e32 vcc, v17, v18", "v_add_u32_e32 v%d, v17, v18" % (48 + k % 8), "s_mov_b64 exec, s[30:31]"] for k in range(16)], [])),
    ("16 dep v_mad chain + 16 s_add", sum([["v_mad_u32_u24 v48, v48, v18, v19", "s_add_u32 s%d, s%d, 1" % (40 + k % 8, 40 + k % 8)] for k in range(16)], [])),
    ("32 dep v_mad chain", ["v_mad_u32_u24 v48, v48, v18, v19" for k in range(32)]),
    ("32 dep v_add chain", ["v_add_u32_e32 v48, v48, v18" for k in range(32)]),
]

STEPS = [
    ("step v6 (as shipped)", "v6", 1),
    ("step v6, no s_nop", "v6_nonop", 1),
    ("step v6, flush test removed", "v6_noflush", 1),
    ("step v6, lean scalar side", "v6_lean", 1),
    ("step v6, lean, test / 2 tokens (r05)", "v6_lean2", 1),
    ("step v6, no scalar bookkeeping", "v6_nosalu", 1),
    ("step v6, no emit block", "v6_noemit", 1),
    ("step v6, VALU only", "v6_valu_only", 1),
    ("step v6, SALU only", "v6_salu_only", 1),
    ("TWO streams per wave, interleaved (per token step)", "two_streams", 2),
    ("TWO streams per wave, one after the other (control)", "two_streams_serial", 2),
    ("pair v7 (per token)", "pair", 1),
    ("pair v7, lean scalar (per token)", "pair_lean", 1),
]

HEADER = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
"""

KERNEL = r"""
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_%(name)s(unsigned* gout, unsigned* sink, int v7) {
  __shared__ __attribute__((aligned(4096))) unsigned lds[8 * 1184 + 520];  // 8 x (4 KiB table + 640 B ring), reciprocals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* tab = lds + wave * 1024;
  // 16 symbols of 16 tokens each.  v6: entry = count << 23 | 2 * below (freq 32 of 512); v7: count << 24 | below (16 of 256)
  for (int s = 0; s < 16; s++) tab[s * 64 + lane] = v7 ? (16u << 24 | 16u * s) : (16u << 23 | 32u * s);
  for (int i = threadIdx.x; i < 8 * 160; i += 512) lds[8 * 1024 + i] = 0;
  // reciprocal of the frequency: v6 {ceil(2^36 / 32), (512 - 32) | 4 << 24} at index count; v7 {0, (256 - 16) | 4 << 24}
  for (int c = threadIdx.x; c < 257; c += 512) {
    lds[8 * 1184 + 2 * c] = v7 ? 0u : 0x80000000u;
    lds[8 * 1184 + 2 * c + 1] = v7 ? (240u | 4u << 24) : (480u | 4u << 24);
  }
  __syncthreads();
  unsigned col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(tab) + 4u * lane;
  unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(lds + 8 * 1024 + wave * 160));
  unsigned x0 = v7 ? 0x10000u : 0x8000u;
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  unsigned* gbase = gout + (blockIdx.x * 8 + wave) * 64;
  const unsigned glo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)gbase);
  const unsigned ghi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)gbase >> 32));
  const unsigned shr = (unsigned)__builtin_amdgcn_readfirstlane(v7 ? 21 : 20);
  unsigned ticks, xout;
  asm volatile(
      "v_mov_b32 v0, %%2\n v_mov_b32 v1, %%3\n"
      "s_mov_b32 s50, 0x9E3779B1\n s_mov_b32 s51, 0x85EBCA77\n s_mov_b32 s52, 0xC2B2AE3D\n s_mov_b32 s53, 0x27D4EB2F\n"
      "v_mov_b32 v17, %%4\n v_mul_lo_u32 v18, v17, s50\n v_mul_lo_u32 v19, v18, s51\n v_mul_lo_u32 v20, v19, s52\n"
      "v_mbcnt_lo_u32_b32 v25, -1, 0\n v_mbcnt_hi_u32_b32 v25, -1, v25\n v_lshlrev_b32 v26, 2, v25\n"
      "s_mov_b32 s20, %%5\n s_mov_b32 s21, 0\n s_mov_b32 s22, 0\n s_mov_b64 s[30:31], exec\n"
      "s_mov_b32 s33, 0xf00\n s_mov_b32 s34, s20\n s_add_u32 s35, s20, 256\n s_mov_b32 s6, s20\n"
      "s_mov_b32 s12, %%6\n s_mov_b32 s13, %%8\n s_mov_b64 vcc, 0x5555\n"
      "s_mov_b32 s40, 1\n s_mov_b32 s41, 2\n s_mov_b32 s42, 3\n s_mov_b32 s43, 4\n s_mov_b32 s44, 5\n s_mov_b32 s45, 6\n s_mov_b32 s46, 7\n s_mov_b32 s47, 8\n"
      "v_mov_b32 v48, 1\n v_mov_b32 v49, 1\n v_mov_b32 v50, 1\n v_mov_b32 v51, 1\n v_mov_b32 v52, 1\n v_mov_b32 v53, 1\n v_mov_b32 v54, 1\n v_mov_b32 v55, 1\n"
      // prime the pipelines: entries of symbol 0, their reciprocals
      "ds_read_b32 v2, v0\n ds_read_b32 v3, v0\n ds_read_b32 v4, v0\n ds_read_b32 v5, v0\n ds_read_b32 v6, v0\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_lshrrev_b32 v22, %%7, v2\n"
      "ds_read_b64 v[32:33], v22 offset:%(rtab)d\n ds_read_b64 v[34:35], v22 offset:%(rtab)d\n ds_read_b64 v[36:37], v22 offset:%(rtab)d\n"
      "ds_read_b64 v[38:39], v22 offset:%(rtab)d\n ds_read_b64 v[40:41], v22 offset:%(rtab)d\n"
      // stream B (two-stream variants): its own state, entry and reciprocal pipelines, symbols, ring cursor
      "v_mov_b32 v7, %%3\n v_mov_b32 v8, v2\n v_mov_b32 v9, v2\n v_mov_b32 v10, v2\n v_mov_b32 v11, v2\n v_mov_b32 v12, v2\n"
      "v_mov_b32 v42, v32\n v_mov_b32 v43, v33\n v_mov_b32 v44, v32\n v_mov_b32 v45, v33\n v_mov_b32 v46, v32\n v_mov_b32 v47, v33\n"
      "v_mov_b32 v48, v32\n v_mov_b32 v49, v33\n v_mov_b32 v50, v32\n v_mov_b32 v51, v33\n"
      "v_mul_lo_u32 v13, v20, s51\n v_mul_lo_u32 v14, v13, s52\n v_mul_lo_u32 v15, v14, s53\n v_mul_lo_u32 v16, v15, s50\n"
      "s_mov_b32 s54, s20\n s_mov_b32 s56, 0\n s_mov_b32 s57, 0\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_memtime s[24:25]\n"
      "s_movk_i32 s23, %(iters)d\n"
      "s_waitcnt lgkmcnt(0)\n"
      "1:\n"
      %(body)s
      // new pseudo-random symbols for the next trip
      "v_mul_lo_u32 v17, v20, s50\n v_mul_lo_u32 v18, v17, s51\n v_mul_lo_u32 v19, v18, s52\n v_mul_lo_u32 v20, v19, s53\n"
      "s_sub_u32 s23, s23, 1\n"
      "s_cmp_lg_u32 s23, 0\n"
      "s_cbranch_scc1 1b\n"
      "s_branch 2f\n"
      %(tail)s
      "2:\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "s_memtime s[26:27]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_sub_u32 s24, s26, s24\n s_subb_u32 s25, s27, s25\n"
      "v_mov_b32 %%0, s24\n"
      "v_mov_b32 %%1, v1\n v_add_u32 %%1, %%1, v48\n v_add_u32 %%1, %%1, v49\n"
      : "=v"(ticks), "=v"(xout)
      : "v"(col), "v"(x0), "v"(h), "s"(ring), "s"(glo), "s"(shr), "s"(ghi)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20",
        "v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55",
        "s6","s7","s12","s13","s20","s21","s22","s23","s24","s25","s26","s27","s30","s31","s33","s34","s35","s36","s37",
        "s40","s41","s42","s43","s44","s45","s46","s47","s50","s51","s52","s53","s54","s55","s56","s57","vcc","scc","memory");
  if (lane == 0) gout[(gridDim.x * 8 + blockIdx.x * 8 + wave) * 64] = ticks;
  if (xout == 0x12345) sink[0] = xout;
}
"""


def ident(name):
    return "".join(ch if ch.isalnum() else "_" for ch in name)


def asm_lines(lines):
    return "\n      ".join('"%s\\n"' % ln for ln in lines)


def main():
    out_dir = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    build = "/tmp/issue_model"
    os.makedirs(build, exist_ok=True)
    iters = 256
    src = [HEADER]
    ents = []
    for name, lines in PURE:
        n = "pure_" + ident(name)
        src.append(KERNEL % {"name": n, "iters": iters, "body": asm_lines(lines), "tail": '""', "rtab": RTAB_OFF})
        ents.append((name, n, 32, 0))
    for name, kind, _ in STEPS:  # _ = streams per wave (token steps per unrolled position)
        n = "step_" + ident(kind)
        del TAIL[:]
        body = asm_lines(body_steps(kind))
        src.append(KERNEL % {"name": n, "iters": iters, "body": body, "tail": asm_lines(TAIL) if TAIL else '""', "rtab": RTAB_OFF})
        ents.append((name, n, UNROLL * _, 1 if kind.startswith("pair") else 0))
    src.append("typedef void (*kfn)(unsigned*, unsigned*, int);\n")
    src.append("struct Ent { const char* n; kfn f; int per_trip; int v7; };\nstatic Ent ents[] = {\n")
    for name, n, per, v7 in ents:
        src.append('  {"%s", k_%s, %d, %d},\n' % (name, n, per, v7))
    src.append("};\n")
    src.append(r"""
int main() {
  unsigned* out; unsigned* sink;
  const int maxblocks = 256 * 4;
  (void)hipMalloc(&out, 4ull * maxblocks * 8 * 64 * 2); (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("ns per unit per SIMD = kernel wall time x 1024 SIMDs / (waves x units per wave); unit = one instruction (pure streams) or one TOKEN STEP (steps)\n");
  printf("%-52s %10s %10s %10s %10s | %12s\n", "stream", "ns@2w/SIMD", "ns@4w", "ns@6w", "ns@8w", "cycles@8w(2.4GHz)");
  for (auto& e : ents) {
    double r[4];
    int ws[4] = {1, 2, 3, 4};  // workgroups of 8 waves per CU -> 2, 4, 6, 8 waves per SIMD
    for (int k = 0; k < 4; k++) {
      int blocks = 256 * ws[k];
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.f, dim3(blocks), dim3(512), 0, 0, out, sink, e.v7);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      r[k] = (double)best * 1e6 * 1024.0 / ((double)blocks * 8.0 * @ITERS@.0 * e.per_trip);
    }
    printf("%-52s %10.2f %10.2f %10.2f %10.2f | %12.1f\n", e.n, r[0], r[1], r[2], r[3], r[3] * 2.4);
  }
  hipError_t err = hipDeviceSynchronize();
  printf("status %d\n", (int)err);
  return 0;
}
""".replace("@ITERS@", str(iters)))
    path = os.path.join(build, "issue_model.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    exe = os.path.join(build, "issue_model")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", path, "-o", exe])
    if "--build-only" in sys.argv:
        return
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    os.makedirs(os.path.join(out_dir, "gpurun_out"), exist_ok=True)
    with open(os.path.join(out_dir, "gpurun_out", "issue_model.txt"), "w") as f:
        f.write(res.stdout)


if __name__ == "__main__":
    main()
