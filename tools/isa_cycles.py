#!/usr/bin/env python
"""VALU issue cost of a stretch of gfx950 ISA, by instruction class (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing --cuda-device-only -S \
          -o /tmp/lmc.s lmcache_amd/csrc/lmc_api.hip
    python tools/isa_cycles.py /tmp/lmc.s k_encode_fusedILi2ELi0 --between v_cmpx_ge_u32_sdwa --nth 45

Round 2 learned the hard way that the NUMBER of VALU instructions per token does not predict the coders' time
(33.5 -> 30.8 instructions per token step changed nothing, HISTORY.md "Round 2e"): what the SIMD spends is
issue time, and that differs by class.  The costs below are the measured ones (tools/probes/valu_rates.py on
MI355X, 8 waves per SIMD, ns per wave-instruction per SIMD; 1 cycle = 0.43 ns at the 2.3 GHz the encoder sustains):

    fast    1.05 ns   v_add_u32 / v_sub_u32 / v_and / v_or / v_xor / v_lshrrev_b32 / v_mov_b32 / v_mul_f32 /
                      v_add_f32 (VOP1/VOP2 and their _e64 forms without modifiers)
    normal  1.85 ns   everything else on the VALU: 3-operand VOP3 (v_bfe, v_lshl_add, v_mad, v_fma, v_perm ...),
                      SDWA / DPP forms, v_cvt_*, v_cmp*, v_addc / v_subb, v_mbcnt, v_lshlrev_b32, packed ops,
                      v_readlane, v_mul_lo / v_mul_hi
    trans   3.5 ns    v_rcp / v_rsq / v_sqrt / v_log / v_exp / v_sin / v_cos
    cnd_e32 9.6 ns    v_cndmask_b32_e32 / _sdwa reading VCC written just before (avoid: the _e64 form is 1.9)

--between PATTERN --nth N   the instructions from the N-th line matching PATTERN up to (not including) the next
                            match: one iteration of an unrolled loop.  Without it: the whole kernel.
The kernel is the first whose mangled name contains KERNEL.  LDS / VMEM instructions are counted, not priced.  Round 5
prices the scalar side as well (tools/probes/issue_model.py, profiles/r05_issue_model.md): a scalar instruction adds
0.43 ns to a VALU-bound step of a full SIMD (1.95 ns when it runs alone), s_nop / an idle s_waitcnt 0.8, and a branch
1.3 ns per instruction when it falls through, 2.0 - 3.6 when it is taken (the listing cannot tell: both are shown).
"""
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
        "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mov_b64"}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")
NS = {"fast": 1.05, "normal": 1.85, "trans": 3.5, "cnd_e32": 9.6}


def classify(mn, ops):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    modified = mn.endswith(("_sdwa", "_dpp"))
    if base.startswith(TRANS):
        return "trans"
    if base == "v_cndmask_b32" and not mn.endswith("_e64"):
        return "cnd_e32"
    if base in FAST and not modified:
        return "fast"
    return "normal"


def kernel_lines(path, key):
    out, on = [], False
    for line in open(path):
        if not on:
            if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", line):
                on = True
            continue
        out.append(line.rstrip("\n"))
        if "s_endpgm" in line and ".Lfunc_end" in "".join(out[-1:]):
            break
        if line.startswith(".Lfunc_end"):
            break
    return out


def main(argv):
    if len(argv) < 3:
        raise SystemExit(__doc__)
    path, key = argv[1], argv[2]
    between, nth = None, 1
    i = 3
    while i < len(argv):
        if argv[i] == "--between":
            between = argv[i + 1]; i += 2
        elif argv[i] == "--nth":
            nth = int(argv[i + 1]); i += 2
        else:
            raise SystemExit(__doc__)
    lines = kernel_lines(path, key)
    if not lines:
        raise SystemExit(f"no kernel matching {key!r} in {path}")
    if between:
        hits = [k for k, l in enumerate(lines) if re.search(between, l)]
        if len(hits) < nth + 1:
            raise SystemExit(f"{len(hits)} lines match {between!r}: no stretch number {nth}")
        lines = lines[hits[nth - 1]:hits[nth]]
    counts = {"fast": 0, "normal": 0, "trans": 0, "cnd_e32": 0}
    other = {"lds": 0, "vmem": 0, "salu": 0, "waitcnt": 0, "branch": 0}
    per = {}
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        parts = l.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        if mn.startswith("v_"):
            c = classify(mn, ops)
            counts[c] += 1
            per[mn] = per.get(mn, 0) + 1
        elif mn.startswith("ds_"):
            other["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "scratch_", "flat_")):
            other["vmem"] += 1
        elif mn == "s_waitcnt":
            other["waitcnt"] += 1
        elif mn.startswith("s_cbranch") or mn == "s_branch":
            other["branch"] += 1
        elif mn.startswith("s_"):
            other["salu"] += 1
    n = sum(counts.values())
    ns = sum(counts[c] * NS[c] for c in counts)
    print(f"{key}: {len(lines)} lines" + (f", stretch {nth} between /{between}/" if between else ""))
    print(f"  VALU instructions {n}: " + ", ".join(f"{c} {counts[c]}" for c in counts if counts[c]))
    print(f"  VALU issue time   {ns:.1f} ns per wave = {ns / 0.43:.0f} cycles at 2.3 GHz "
          f"(x 8 waves per SIMD = {8 * ns:.0f} ns per token step of a full SIMD)")
    print("  other: " + ", ".join(f"{k} {v}" for k, v in other.items() if v))
    salu_ns = other["salu"] * 0.43 + other["branch"] * 0.43
    print(f"  scalar side beside the VALU stream: {other['salu'] + other['branch']} instructions = +{salu_ns:.1f} ns per step "
          f"(if every branch is taken: +{salu_ns + other['branch'] * 1.5:.1f}); issue floor of the stretch "
          f"{ns + salu_ns:.1f} ns per wave-step")
    print("  " + " ".join(f"{m}:{c}" for m, c in sorted(per.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main(sys.argv)
