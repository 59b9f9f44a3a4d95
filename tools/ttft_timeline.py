"""rocprofv3 --kernel-trace CSV of tools/probes/ttft_hbm_tier.py -> the timeline of its LAST round of schedules.

    python tools/ttft_timeline.py out/t_kernel_trace.csv [--episodes 8] [--md]

An episode = one warm prefix + one model step: the k_decode launches of the retrieve (D) and the 32 GEMVs of the proxy
step (M), separated from the next episode by > 0.2 ms of idle GPU.  Per episode: the decode launches with their
durations, how long D and M ran side by side, and what each cost while they did.
"""
import argparse
import csv


def load(path):
    ev = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        k = "D" if "k_decode" in n else ("M" if n.startswith("Cijk") else None)
        if k:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    ev.sort()
    eps = [[ev[0]]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in eps[-1]) > 200_000:
            eps.append([e])
        else:
            eps[-1].append(e)
    return [ep for ep in eps if sum(1 for e in ep if e[2] == "M") == 32]


def overlap(a, b):
    return max(0, min(a[1], b[1]) - max(a[0], b[0]))


def describe(ep):
    t0 = ep[0][0]
    D = [e for e in ep if e[2] == "D"]
    M = [e for e in ep if e[2] == "M"]
    total = (max(e[1] for e in ep) - t0) / 1e3
    first_m = (M[0][0] - t0) / 1e3
    d_end = (max(e[1] for e in D) - t0) / 1e3
    both = sum(overlap(d, m) for d in D for m in M) / 1e3
    m_over = [m for m in M if any(overlap(d, m) > 0.5 * (m[1] - m[0]) for d in D)]
    m_free = [m for m in M if not any(overlap(d, m) > 0 for d in D)]
    dur = lambda L: sum(e[1] - e[0] for e in L) / 1e3 / max(1, len(L))
    return {"decode_launches": len(D), "total_us": round(total, 1), "first_gemv_starts_us": round(first_m, 1),
            "last_decode_ends_us": round(d_end, 1), "decode_busy_us": round(sum(e[1] - e[0] for e in D) / 1e3, 1),
            "side_by_side_us": round(both, 1), "gemv_us_beside_decode": round(dur(m_over), 1), "gemvs_beside_decode": len(m_over),
            "gemv_us_alone": round(dur(m_free), 1), "gemvs_alone": len(m_free),
            "decode_us_each": [round((e[1] - e[0]) / 1e3) for e in D]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--episodes", type=int, default=8, help="how many of the last episodes (one per schedule of the last round)")
    ap.add_argument("--md", action="store_true")
    a = ap.parse_args()
    eps = load(a.csv)[-a.episodes:]
    rows = [describe(ep) for ep in eps]
    if a.md:
        print("| decode launches | total (first kernel -> last), us | first GEMV starts | last decode ends | decode busy | D and M side by side | GEMV beside a decode (n) | GEMV alone (n) | decode launches, us each |")
        print("|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            print(f"| {r['decode_launches']} | {r['total_us']} | {r['first_gemv_starts_us']} | {r['last_decode_ends_us']} | {r['decode_busy_us']} | "
                  f"{r['side_by_side_us']} | {r['gemv_us_beside_decode']} ({r['gemvs_beside_decode']}) | {r['gemv_us_alone']} ({r['gemvs_alone']}) | "
                  f"{' '.join(str(x) for x in r['decode_us_each'])} |")
    else:
        for r in rows:
            print(r)


if __name__ == "__main__":
    main()
