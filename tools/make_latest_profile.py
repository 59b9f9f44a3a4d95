#!/usr/bin/env python
"""profiles/latest.json from the rocprofv3 PMC passes of one round (what bench.py reads for `roofline.traffic` and
the VALU roof -- so those numbers are the committed profile's, never hand-copied literals).

    python tools/make_latest_profile.py <tag> <sq.db> <fetch.db> <write.db>  > profiles/latest.json

sq.db     --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES
fetch.db  --pmc FETCH_SIZE GRBM_GUI_ACTIVE
write.db  --pmc WRITE_SIZE
(separate passes, as MI355X_MICROARCH.md prescribes).  Only full-context dispatches (>= 2 M work-items) count.
The byte counters are corrected with MEASURED factors: profiles/r04_counter_calibration.json (tools/probes/
counter_calibration under the same two PMC passes: 1 GiB moved by each of the kernels' access patterns) -- FETCH_SIZE
reports half of every read pattern the kernels use (16 B and 4 B per lane, temporal or not, global_load_lds alike),
WRITE_SIZE is exact.  (Rounds 1-3 doubled FETCH_SIZE for the 16 B / lane loads only and took the 4 B / lane symbol
loads as reported: their `traffic` was low by the symbols' re-reads.)  The totals (`traffic_bytes_per_step`,
`valu_insts_per_step`) are those of the DEFAULT launch path: k_encode_fused when it ran, else k_quantize + k_cdf_encode.
"""
import json
import sqlite3
import sys

import os

MIN_GRID = 2_000_000  # k_encode_fused: 4096 workgroups x 512
CAL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r04_counter_calibration.json")))["factors"]
# kernel name prefix -> (FETCH_SIZE factor, WRITE_SIZE factor) from the calibrated patterns the kernel uses
KERNELS = {"k_encode_fused": (max(CAL["rd16_nt"], CAL["rd4"], CAL["rd4_nt"]), CAL["wr4_nt"]),
           "k_quantize": (CAL["rd16"], CAL["wr16"]),
           "k_cdf_encode": (max(CAL["rd4"], CAL["rd_lds16"]), CAL["wr16_nt"]),
           "k_decode": (CAL["rd4_nt"], CAL["wr2_buf"])}
assert min(CAL[k] for k in ("rd16_nt", "rd4", "rd4_nt", "rd16", "rd_lds16")) > 0.97 * max(CAL[k] for k in ("rd16_nt", "rd4", "rd4_nt", "rd16", "rd_lds16")), \
    "the read patterns no longer share one FETCH_SIZE factor: split the kernels' reads by pattern"


# Share of a kernel's DYNAMIC VALU instructions that are of the fast class (v_add / sub / and / or / xor / lshrrev / mov /
# mul_f32 / add_f32 without modifiers: 1.05 ns per wave-instruction per SIMD against 1.85 for the rest; tools/probes/
# valu_rates.py): the static class mix of the kernel's hot loops (tools/isa_cycles.py --between ...) weighted by what
# each loop contributes per token step -- profiles/r05_issue_model.md has the table.
FAST_SHARE = {"k_encode_fused": 0.27, "k_cdf_encode": 0.24, "k_decode": 0.48, "k_quantize": 0.37}


def counters(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*) from counters_collection "
                      "where grid_size >= ? group by kernel_name, counter_name", (MIN_GRID,)).fetchall()
    out = {}
    for name, cn, v, dur, n in rows:
        for k in KERNELS:
            if k in name and "<" in name:
                out.setdefault(k, {})[cn] = (v, dur, n)
    return out


def main(argv):
    tag, sq, fe, wr = argv[1:5]
    csq, cfe, cwr = counters(sq), counters(fe), counters(wr)
    res = {"source": [f"profiles/{tag}_pmc.md", f"profiles/{tag}_bench_kernel_stats.md", "profiles/r04_counter_calibration.md"],
           "kernels": {}}
    traffic = valu = 0.0
    dominant, dom_us = None, 0.0
    for k, (factor, wfactor) in KERNELS.items():
        if k not in csq or k not in cfe or k not in cwr:
            continue
        fetch_kib = cfe[k]["FETCH_SIZE"][0]
        write_kib = cwr[k]["WRITE_SIZE"][0]
        gui, dur_ns, _ = cfe[k]["GRBM_GUI_ACTIVE"]
        clock_ghz = gui / 8.0 / dur_ns  # 8 XCDs tick the counter
        insts = csq[k]["SQ_INSTS_VALU"][0]
        active, sq_ns, _ = csq[k]["SQ_ACTIVE_INST_VALU"]
        busy = active * 4.0 / (1024.0 * sq_ns * clock_ghz)
        hbm = (fetch_kib * factor + write_kib * wfactor) * 1024.0
        res["kernels"][k] = {"avg_us_profiled": round(sq_ns / 1e3, 1), "hbm_bytes": int(hbm),
                             "fetch_bytes": int(fetch_kib * factor * 1024.0), "write_bytes": int(write_kib * wfactor * 1024.0),
                             "valu_insts": int(insts), "salu_insts": int(csq[k]["SQ_INSTS_SALU"][0]),
                             "lds_insts": int(csq[k]["SQ_INSTS_LDS"][0]), "valu_fast_share": FAST_SHARE[k],
                             "wait_any_frac": round(csq[k]["SQ_WAIT_ANY"][0] / csq[k]["SQ_WAVE_CYCLES"][0], 3),
                             "valu_busy": round(busy, 3), "clock_GHz": round(clock_ghz, 3)}
        default_path = k == "k_encode_fused" or (k != "k_decode" and "k_encode_fused" not in csq)
        if default_path:
            traffic += hbm
            valu += insts
            if sq_ns > dom_us:
                dominant, dom_us = k, sq_ns
    res["traffic_bytes_per_step"] = int(traffic)
    res["valu_insts_per_step"] = int(valu)
    res["dominant_kernel"] = dominant
    res["valu_busy_dominant_kernel"] = res["kernels"][dominant]["valu_busy"]
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv)
