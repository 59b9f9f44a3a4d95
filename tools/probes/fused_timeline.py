"""Reads gpurun_out/fused_timeline.bin (tools/probes/fused_timeline.hip) and prints how many work items of one
k_encode_fused launch are in which phase over time, plus per-phase duration statistics by generation."""
import sys
import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fused_timeline.bin"
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
n = t.shape[0]
ts = t[:, :6].astype(np.int64)
t0 = ts[:, 0].min()
us = (ts - t0) / 100.0  # 100 MHz ticks -> microseconds
end = us[:, 5].max()
print(f"{n} items; launch span {end:.1f} us (first start to last end)")
names = ["wait+phase A", "pass 1", "look-back", "pass 2 (wave 0)", "pass 2 (rest)"]
dur = np.diff(us, axis=1)
order = np.argsort(us[:, 0])
print("\nphase durations (us) by start order (quartiles of the launch's items)")
print("%-18s" % "items" + "".join("%18s" % nm for nm in names) + "%12s" % "total")
for q in range(8):
    sel = order[q * n // 8:(q + 1) * n // 8]
    print("%-18s" % f"{q * n // 8}..{(q + 1) * n // 8 - 1}" + "".join("%18.1f" % dur[sel, k].mean() for k in range(5)) + "%12.1f" % (us[sel, 5] - us[sel, 0]).mean())
print("\nitems in each phase over time (every %d us)" % max(1, int(end / 40)))
step = max(1.0, end / 40)
print("%8s %8s %8s %8s %8s %8s %8s" % ("t us", "resident", "phase A", "pass 1", "lookback", "pass 2", "started"))
for k in range(int(end / step) + 1):
    x = k * step
    res = ((us[:, 0] <= x) & (us[:, 5] > x)).sum()
    a = ((us[:, 0] <= x) & (us[:, 1] > x)).sum()
    p1 = ((us[:, 1] <= x) & (us[:, 2] > x)).sum()
    lb = ((us[:, 2] <= x) & (us[:, 3] > x)).sum()
    p2 = ((us[:, 3] <= x) & (us[:, 5] > x)).sum()
    print("%8.1f %8d %8d %8d %8d %8d %8d" % (x, res, a, p1, lb, p2, (us[:, 0] <= x).sum()))
hw = t[:, 6]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw.astype(np.int64) >> 8) & 0xf
se = (hw.astype(np.int64) >> 13) & 0x7
print("\nitems per XCC:", np.bincount(xcc, minlength=8).tolist())
first = order[: min(n, 1024)]
print("first 1024 starts: per XCC", np.bincount(xcc[first], minlength=8).tolist(), " start spread %.1f us" % (us[first, 0].max() - us[first, 0].min()))
