"""Achievable HBM streaming bandwidth on this device (read+write), for sizing roofline fractions."""
import torch, time
dev = torch.device("cuda:0")
for mb in (512, 2048):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"copy {mb} MiB: {ms:.3f} ms  read+write {2 * n * 2 / ms / 1e9:.2f} TB/s")
    # read-only: sum
    for _ in range(3):
        s = x.view(torch.int16).sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        s = x.view(torch.int32).sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"sum  {mb} MiB: {ms:.3f} ms  read {n * 2 / ms / 1e9:.2f} TB/s")
    # write-only: fill
    e0.record()
    for _ in range(20):
        y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"fill {mb} MiB: {ms:.3f} ms  write {n * 2 / ms / 1e9:.2f} TB/s")
