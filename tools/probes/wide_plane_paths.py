"""C = 2048 and C = 4096 (BASELINE configs[0] shape, fp16) at 16 .. 128 chunks: the two-kernel path, the fused kernel, and what AUTO picks."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lmcache_amd import native
from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
ctx = native.get_context(0)
L, D, dt = 32, 128, torch.float16
bins = CacheGenConfig.from_model_name("mistralai/Mistral-7B-Instruct-v0.2").plane_bins(L)
for H, T in ((16, 4096), (16, 16384), (16, 32768), (32, 4096), (32, 16384)):
    cs = 256
    kv = tuple((torch.rand(T, H, D, device=dev).to(dt), torch.rand(T, H, D, device=dev).to(dt)) for _ in range(L))
    lay = native.KVLayout.from_kv_tuple(kv, "vllm")
    n = T // cs
    stride = native.r16(native.blob_bound(L, cs, H, D))
    blobs = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int32, device=dev)
    raw = L * 2 * T * H * D * 2
    for path in ("two_kernels", "fused", "auto"):
        ctx.set_encode_path(path)
        def enc():
            ctx.encode_chunks(lay, 0, T, cs, bins, blobs.data_ptr(), stride, sizes.data_ptr())
        for _ in range(30): enc()
        torch.cuda.synchronize(); ctx.raise_on_status("enc")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): enc()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"C={H*D} T={T} {path}: {t:.3f} ms = {raw/t/1e6:.0f} GB/s", flush=True)
    del kv, blobs
