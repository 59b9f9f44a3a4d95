#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats / --pmc) as markdown tables.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/<name>.md
    python tools/rocpd_stats.py db [--min-grid N] [--per UNITS]

--min-grid N   PMC table: only dispatches with at least N work-items (drops the small launches of the
               pipelined retrieve leg so that the averages are those of the full-context jobs)
--per UNITS    also print every counter divided by UNITS (e.g. wave token-steps per dispatch)
"""
import sqlite3
import sys


def main(argv):
    path = argv[1]
    min_grid, per = 0, None
    i = 2
    while i < len(argv):
        if argv[i] == "--min-grid":
            min_grid = int(argv[i + 1]); i += 2
        elif argv[i] == "--per":
            per = float(argv[i + 1]); i += 2
        else:
            raise SystemExit(__doc__)
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
    if min_grid:
        rows = db.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
            "where grid_x * grid_y * grid_z >= ? group by name order by sum(duration) desc", (min_grid,)).fetchall()
        print(f"\n## Kernel dispatches of >= {min_grid} work-items only\n")
        print("| kernel | calls | total ms | avg us | min us | max us |")
        print("|---|---|---|---|---|---|")
        for name, n, tot, avg, mn, mx in rows:
            print(f"| `{name[:110]}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
    try:
        pmc = db.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*) "
                         "from counters_collection where grid_size >= ? group by kernel_name, counter_name "
                         "order by kernel_name, counter_name", (min_grid,)).fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        extra = f", dispatches of >= {min_grid} work-items" if min_grid else ""
        print(f"\n## PMC counters (average per dispatch{extra}; FETCH_SIZE / WRITE_SIZE in KiB)\n")
        hdr = "| kernel | counter | avg per dispatch | avg dispatch us | dispatches |"
        sep = "|---|---|---|---|---|"
        if per:
            hdr += f" per unit (/{per:.6g}) |"
            sep += "---|"
        print(hdr + "\n" + sep)
        for name, cn, v, dur, n in pmc:
            if name.startswith("void at::") or name.startswith("__amd"):
                continue
            line = f"| `{name[:60]}` | {cn} | {v:.5g} | {dur / 1e3:.1f} | {n} |"
            if per:
                line += f" {v / per:.3f} |"
            print(line)


if __name__ == "__main__":
    main(sys.argv)
