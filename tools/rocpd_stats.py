#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) as the
per-kernel table rocprofv3 would print in CSV mode.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/<name>.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{name[:110]}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
              f"{100.0 * tot / total:.1f} |")
    try:
        pmc = db.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*) "
                         "from counters_collection group by kernel_name, counter_name "
                         "order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n## PMC counters (average per dispatch; FETCH_SIZE / WRITE_SIZE in KiB)\n")
        print("| kernel | counter | avg per dispatch | avg dispatch us | dispatches |\n|---|---|---|---|---|")
        for name, cn, v, dur, n in pmc:
            if name.startswith("void at::") or name.startswith("__amd"):
                continue
            print(f"| `{name[:60]}` | {cn} | {v:.5g} | {dur / 1e3:.1f} | {n} |")


if __name__ == "__main__":
    main(sys.argv[1])
