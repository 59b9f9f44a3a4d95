#!/usr/bin/env python
"""profiles/<round>_counter_calibration.md (+ .json) from the two PMC passes of tools/probes/counter_calibration:

    python tools/counter_calibration.py <fetch.db> <write.db> <out.md> <out.json>

factor = bytes really moved / bytes the counter reports (FETCH_SIZE, WRITE_SIZE are in KiB): what a reading of that
access pattern has to be multiplied by.  tools/make_latest_profile.py reads the .json."""
import json
import sqlite3
import sys

BYTES = 1 << 30
PATTERNS = {
    "rd16_nt": ("FETCH_SIZE", "16 B / lane non-temporal loads (raw KV in quantize_oct_fused)"),
    "rd16": ("FETCH_SIZE", "16 B / lane loads (k_quantize)"),
    "rd4": ("FETCH_SIZE", "4 B / lane loads, 256 B per wave (symbol workspace, decoder stream words)"),
    "rd4_nt": ("FETCH_SIZE", "4 B / lane non-temporal loads (last-use symbol loads)"),
    "rd_lds16": ("FETCH_SIZE", "global_load_lds_dwordx4 (copy_stream16)"),
    "wr16": ("WRITE_SIZE", "16 B / lane stores (symbol workspace)"),
    "wr16_nt": ("WRITE_SIZE", "16 B / lane non-temporal stores (placed streams, two-kernel path)"),
    "wr4_nt": ("WRITE_SIZE", "4 B / lane non-temporal stores, 256 B per wave (the coder's pieces)"),
    "wr2_buf": ("WRITE_SIZE", "2 B / lane raw buffer stores, 128 B per wave, non-temporal (decoder output rows)"),
}


def read(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, avg(value), avg(duration), count(*) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {name.split("(")[0].split()[-1]: (v, dur, n) for name, v, dur, n in rows}


def main(argv):
    fe, wr, out_md, out_json = argv[1:5]
    got = {"FETCH_SIZE": read(fe, "FETCH_SIZE"), "WRITE_SIZE": read(wr, "WRITE_SIZE")}
    factors = {}
    lines = ["# Counter calibration: what FETCH_SIZE / WRITE_SIZE report for the product's access patterns", "",
             "`tools/probes/counter_calibration` under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
             "(two passes): every kernel moves exactly 1 GiB (four times the Infinity Cache) with one access pattern of the "
             "encoder / decoder.  factor = bytes moved / bytes reported: what `tools/make_latest_profile.py` multiplies a "
             "reading of that pattern by (MI355X_MICROARCH.md gives 2.0 for the 16 B / lane streaming read and calls every "
             "other width, and WRITE_SIZE, uncalibrated).", "",
             "| kernel | pattern | counter | reported MiB | moved MiB | factor | avg us | GB/s |", "|---|---|---|---|---|---|---|---|"]
    for k, (counter, what) in PATTERNS.items():
        if k not in got[counter]:
            continue
        kib, dur, n = got[counter][k]
        f = BYTES / (kib * 1024.0) if kib else float("nan")
        factors[k] = round(f, 4)
        lines.append(f"| `{k}` | {what} | {counter} | {kib / 1024:.1f} | {BYTES / 2**20:.0f} | {f:.3f} | {dur / 1e3:.1f} | "
                     f"{BYTES / dur:.0f} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump({"bytes": BYTES, "factors": factors}, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv)
