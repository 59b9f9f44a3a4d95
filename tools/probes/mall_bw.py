import torch
dev = torch.device("cuda:0")
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 2048):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(20, 4096 // mb)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy src {mb} MiB + dst {mb} MiB: {ms*1000:.1f} us  read+write {2 * n * 2 / ms / 1e9:.2f} TB/s", flush=True)
