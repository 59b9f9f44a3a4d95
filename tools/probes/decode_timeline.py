"""Reads <fused_timeline.bin>.decode (tools/probes/fused_timeline.hip, a -DLMC_EXP_TIMELINE library): per k_decode wave
{start, prologue done, token loops done, nsym << 48 | hw id} in 100 MHz ticks.  Prints the waves' lifetimes by kind of
plane and how many waves are resident / in their prologue over time."""
import sys
import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fused_timeline.bin.decode"
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
ts = t[:, :3].astype(np.int64)
ok = ts[:, 2] > 0
t, ts = t[ok], ts[ok]
us = (ts - ts[:, 0].min()) / 100.0
nsym = (t[:, 3] >> np.uint64(48)).astype(np.int64)
end = us[:, 2].max()
print(f"{len(us)} waves; launch span {end:.1f} us")
for name, sel in (("<= 16 symbols", nsym <= 16), ("> 16 symbols", nsym > 16)):
    if sel.any():
        pro, loop = us[sel, 1] - us[sel, 0], us[sel, 2] - us[sel, 1]
        print(f"{name:14s} {sel.sum():6d} waves: prologue {pro.mean():6.1f} us (p10 {np.percentile(pro, 10):.1f}, p90 {np.percentile(pro, 90):.1f}), "
              f"token loops {loop.mean():6.1f} us (p10 {np.percentile(loop, 10):.1f}, p90 {np.percentile(loop, 90):.1f})")
step = max(1.0, end / 40)
print("%8s %9s %9s %9s %9s" % ("t us", "resident", "prologue", "in loops", "started"))
for k in range(int(end / step) + 1):
    x = k * step
    res = ((us[:, 0] <= x) & (us[:, 2] > x)).sum()
    pro = ((us[:, 0] <= x) & (us[:, 1] > x)).sum()
    print("%8.1f %9d %9d %9d %9d" % (x, res, pro, res - pro, (us[:, 0] <= x).sum()))
