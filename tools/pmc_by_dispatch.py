#!/usr/bin/env python
"""Per-dispatch counters of one kernel from a rocprofv3 --pmc database, in dispatch order, next to the dispatch's
duration -- to see which counter moves when the same kernel runs fast or slow (tools/probes/encode_modes).

    python tools/pmc_by_dispatch.py <results.db> <kernel name substring>
"""
import sqlite3
import sys


def main(argv):
    db = sqlite3.connect(argv[1])
    rows = db.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like ? "
                      "order by dispatch_id", (f"%{argv[2]}%",)).fetchall()
    names = sorted({r[1] for r in rows})
    by = {}
    for d, n, v, dur in rows:
        by.setdefault(d, {"us": dur / 1e3})[n] = v
    print("| dispatch | us | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for d in sorted(by):
        print(f"| {d} | {by[d]['us']:.1f} | " + " | ".join(f"{by[d].get(n, 0):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main(sys.argv)
