#!/usr/bin/env python
"""profiles/latest.json from the rocprofv3 PMC passes of one round (what bench.py reads for `roofline.traffic` and
the VALU roof -- so those numbers are the committed profile's, never hand-copied literals).

    python tools/make_latest_profile.py <tag> <sq.db> <fetch.db> <write.db>  > profiles/latest.json

sq.db     --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES
fetch.db  --pmc FETCH_SIZE GRBM_GUI_ACTIVE
write.db  --pmc WRITE_SIZE
(separate passes, as MI355X_MICROARCH.md prescribes).  Only full-context dispatches (>= 4 M work-items) count.
FETCH_SIZE is doubled for k_quantize (16 B / lane streaming loads: the guide's gfx950 correction), taken as is for
k_cdf_encode (4 B / lane symbol loads and its own L2-hot scratch re-read).  k_encode_fused (the default launch at
the bench's size) mixes both kinds: its raw-KV loads are the quantiser's (16 B / lane, tallied at half), the rest is
the coder's -- and its FETCH_SIZE is indeed k_quantize's + k_cdf_encode's to 3 % (the symbols do NOT come back from
L2: 128 plane-chunks in flight per XCD are 16-32 MB against 4 MB of L2) -- so the half the counter misses is added
once: FETCH_SIZE(k_encode_fused) + FETCH_SIZE(k_quantize) when the database holds both (the A/B probe), x 1.55
otherwise.  The totals (`traffic_bytes_per_step`, `valu_insts_per_step`) are those of the DEFAULT launch path:
k_encode_fused when it ran, else k_quantize + k_cdf_encode.
"""
import json
import sqlite3
import sys

MIN_GRID = 2_000_000  # k_encode_fused: 4096 workgroups x 512
KERNELS = {"k_encode_fused": 1.55, "k_quantize": 2.0, "k_cdf_encode": 1.0}  # kernel name prefix -> FETCH_SIZE factor


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
    res = {"source": [f"profiles/{tag}_pmc.md", f"profiles/{tag}_bench_kernel_stats.md"], "kernels": {}}
    traffic = valu = 0.0
    dominant, dom_us = None, 0.0
    for k, factor in KERNELS.items():
        if k not in csq or k not in cfe or k not in cwr:
            continue
        fetch_kib = cfe[k]["FETCH_SIZE"][0]
        write_kib = cwr[k]["WRITE_SIZE"][0]
        gui, dur_ns, _ = cfe[k]["GRBM_GUI_ACTIVE"]
        clock_ghz = gui / 8.0 / dur_ns  # 8 XCDs tick the counter
        insts = csq[k]["SQ_INSTS_VALU"][0]
        active, sq_ns, _ = csq[k]["SQ_ACTIVE_INST_VALU"]
        busy = active * 4.0 / (1024.0 * sq_ns * clock_ghz)
        hbm = (fetch_kib * factor + write_kib) * 1024.0
        if k == "k_encode_fused" and "k_quantize" in cfe:
            hbm = (fetch_kib + cfe["k_quantize"]["FETCH_SIZE"][0] + write_kib) * 1024.0
        res["kernels"][k] = {"avg_us_profiled": round(sq_ns / 1e3, 1), "hbm_bytes": int(hbm), "valu_insts": int(insts),
                             "valu_busy": round(busy, 3), "clock_GHz": round(clock_ghz, 3)}
        default_path = k == "k_encode_fused" or "k_encode_fused" not in csq
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
