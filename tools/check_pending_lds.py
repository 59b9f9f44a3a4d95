#!/usr/bin/env python
"""Static check of hand-placed LDS reads (no GPU needed).

The counts coder (k_encode_counts.h, LMC_COUNTS_LDSASM) issues ds_read instructions from inside asm blocks and waits
for them at the top of the NEXT block; the compiler knows nothing of either.  This script walks the compiled ISA of a
kernel and reports any instruction that touches a VGPR whose ds_read (issued inside an asm block) has not been
followed by an `s_waitcnt lgkmcnt(0)` yet -- e.g. a register copy the allocator placed at a loop back edge.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing --cuda-device-only -S \
          -o /tmp/lmc.s lmcache_amd/csrc/lmc_api.hip
    python tools/check_pending_lds.py /tmp/lmc.s k_encode_fused k_cdf_encodeILb1ELb1
Straight-line scan in layout order (branches are not followed: an order-of-text check).

KNOWN FLAW, and why the shipped coder no longer needs this script: a compiler-emitted `s_waitcnt lgkmcnt(0)` inside a
branch that is normally SKIPPED (the ring flush) clears the pending set here although the hardware never executes it --
which is how a `v_mov` of a still-pending ds_read_b64 half got past this check and corrupted streams at full size
(round 3).  The coder now lets the compiler issue and track those loads and only pins their
position; the script stays for whoever hides an LDS read in an asm block again.
"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(lines, name):
    pending = {}  # vgpr -> line number of its ds_read
    in_asm = False
    bad = 0
    for n, line in enumerate(lines, 1):
        t = line.split(";")[0].strip() if not line.strip().startswith(";;#") else line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        parts = t.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if mn == "s_waitcnt":
            if "lgkmcnt(0)" in t:
                pending.clear()
            continue
        touched = set()
        for o in ops:
            touched |= regs(o)
        hit = touched & set(pending)
        if hit:
            bad += 1
            print(f"{name}: line {n}: `{t}` touches v{sorted(hit)} pending since line {[pending[h] for h in sorted(hit)]}")
        if in_asm and mn.startswith("ds_read"):
            for r in regs(ops[0]):
                pending[r] = n
    return bad


def main():
    path, keys = sys.argv[1], sys.argv[2:]
    text = open(path).read().split("\n")
    total = 0
    for key in keys:
        out, on, nm = [], False, None
        for line in text:
            m = re.match(r"^(_Z\w*" + re.escape(key) + r"\w*):", line)
            if m and not on:
                on, nm = True, m.group(1)
                out = []
                continue
            if on:
                out.append(line)
                if line.startswith(".Lfunc_end"):
                    total += scan(out, nm)
                    print(f"{nm}: scanned {len(out)} lines")
                    on = False
    print("pending-register violations:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
