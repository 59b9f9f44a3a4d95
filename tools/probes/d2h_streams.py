import sys, time, json, torch
sys.path.insert(0, '.')
import bench
from lmcache_amd import native
from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
kv = bench.make_kv(dev, 0)
layout = native.KVLayout.from_kv_tuple(kv, "vllm")
bins = bench.cachegen_bins_llama8b()
codec = get_codec(0)
raw = bench.L*2*bench.CTX*bench.H*bench.D*2
for n in (1, 2):
    codec.d2h_streams = n
    arena = PinnedArena(slab_bytes=620 << 20)
    def once():
        arena.reset()
        job = codec.encode(layout, 0, bench.CTX, bench.CHUNK, bins)
        sz = codec.sizes_of(job)
        hb, done = codec.offload(job, sz, arena)
        done.synchronize()
        return sum(sz)
    once()
    t0 = time.perf_counter()
    for _ in range(5):
        nb = once()
    dt = (time.perf_counter() - t0) / 5
    print(n, "streams: ms", round(dt*1e3, 2), "blob GB/s", round(nb/dt/1e9, 1), "raw GB/s", round(raw/dt/1e9, 1))
    arena.close()
