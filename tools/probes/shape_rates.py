"""Encode / decode rates of other geometries than the headline one (HBM-resident, one job each)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lmcache_amd import native
from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
ctx = native.get_context(0)
CASES = [("Llama-3-8B GQA  C=1024 bf16", 32, 8, 128, torch.bfloat16, 4096, "Llama-3-8B"),
         ("7B MHA        C=4096 fp16", 32, 32, 128, torch.float16, 2048, "mistralai/Mistral-7B-Instruct-v0.2"),
         ("70B TP8 rank   C=128  bf16", 80, 1, 128, torch.bfloat16, 8192, "Llama-3-70B"),
         ("C=512 bf16", 32, 4, 128, torch.bfloat16, 4096, "Llama-3-8B")]
for name, L, H, D, dt, T, model in CASES:
    cs = 256
    kv = tuple((torch.rand(T, H, D, device=dev).to(dt), torch.rand(T, H, D, device=dev).to(dt)) for _ in range(L))
    lay = native.KVLayout.from_kv_tuple(kv, "vllm")
    bins = CacheGenConfig.from_model_name(model).plane_bins(L)
    n = T // cs
    stride = native.r16(native.blob_bound(L, cs, H, D))
    blobs = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int32, device=dev)
    raw = L * 2 * T * H * D * 2
    def enc():
        ctx.encode_chunks(lay, 0, T, cs, bins, blobs.data_ptr(), stride, sizes.data_ptr())
    enc(); torch.cuda.synchronize(); ctx.raise_on_status("enc")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): enc()
    e1.record(); torch.cuda.synchronize()
    tenc = e0.elapsed_time(e1) / 5
    out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    ol = native.KVLayout.from_kv_tuple(out, "vllm")
    ctx.decode_chunks(blobs.data_ptr(), stride, n, ol, 0, cs); torch.cuda.synchronize(); ctx.raise_on_status("dec")
    e0.record()
    for _ in range(5): ctx.decode_chunks(blobs.data_ptr(), stride, n, ol, 0, cs)
    e1.record(); torch.cuda.synchronize()
    tdec = e0.elapsed_time(e1) / 5
    print(f"{name}: {raw/1e6:.0f} MB raw, encode {tenc:.3f} ms = {raw/tenc/1e6:.0f} GB/s, decode {tdec:.3f} ms = {raw/tdec/1e6:.0f} GB/s, ratio {raw/int(sizes.sum()):.2f}x", flush=True)
    del kv, out, blobs
