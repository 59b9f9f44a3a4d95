"""Where the host time of an HBM-tier retrieve_layerwise goes (cProfile over 30 calls per schedule).

    python tools/probes/ttft_host_profile.py [layers_per_launch ...]      (default: 1 2 32)
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from lmcache_amd.cache_engine import LMCacheEngine
    from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
    kv = bench.make_kv(dev, 0, "rand")
    meta = LMCacheEngineMetadata(bench.MODEL, 1, 0, "vllm", "bfloat16")
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=bench.CHUNK, backend="cuda", local_serde="cachegen")
    engine = LMCacheEngine(cfg, meta)
    toks = torch.randint(0, 32000, (bench.CTX,), generator=torch.Generator().manual_seed(7))
    engine.store(toks, kv)
    side = torch.cuda.Stream(device=dev, priority=-1)
    # a range size, or a schedule of range sizes "4,28"
    scheds = [(int(a) if "," not in a else tuple(int(x) for x in a.split(","))) for a in sys.argv[1:]] or [1, 2, 32]
    for lpl in scheds:
        def once():
            with torch.cuda.stream(side):
                res = engine.retrieve_layerwise(toks, layers_per_launch=lpl)
            return res
        for _ in range(3):
            once().finish()
            torch.cuda.synchronize()
        plain = []
        for _ in range(30):  # without the profiler: what bench.py's host_ms_before_the_model_can_start measures
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = once()
            plain.append((time.perf_counter() - t0) * 1e3)
            res.finish()
        print(f"== layers_per_launch {lpl}: median {sorted(plain)[15]:.3f} ms per call, min {min(plain):.3f} (no profiler)")
        ts = []
        pr = cProfile.Profile()
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr.enable()
            res = once()
            pr.disable()
            ts.append((time.perf_counter() - t0) * 1e3)
            res.finish()
        print(f"== layers_per_launch {lpl}: median {sorted(ts)[15]:.3f} ms per call (with the profiler's overhead)")
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(28)
    engine.close()


if __name__ == "__main__":
    main()
