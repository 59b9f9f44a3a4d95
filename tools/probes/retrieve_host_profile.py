"""Probe: where the host time of a warm-prefix retrieve goes (cProfile of engine.retrieve_layerwise / retrieve on the
HBM-resident encoded tier; the GPU work is asynchronous, so this is the latency in front of the first launch)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lmcache_amd.cache_engine import LMCacheEngine  # noqa: E402
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata  # noqa: E402

dev = torch.device("cuda:0")
kv = bench.make_kv(dev, 0, "rand")
meta = LMCacheEngineMetadata(bench.MODEL, 1, 0, "vllm", "bfloat16")
engine = LMCacheEngine(LMCacheEngineConfig.from_legacy(chunk_size=bench.CHUNK, backend="cuda", local_serde="cachegen"), meta)
toks = torch.randint(0, 32000, (bench.CTX,), generator=torch.Generator().manual_seed(7))
engine.store(toks, kv)
for _ in range(3):
    engine.retrieve_layerwise(toks, layers_per_launch=(2, 6, 24)).finish()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = engine.retrieve_layerwise(toks, layers_per_launch=(2, 6, 24))
    ts.append((time.perf_counter() - t0) * 1e3)
    r.finish()
print("retrieve_layerwise host ms:", sorted(ts))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    r = engine.retrieve_layerwise(toks, layers_per_launch=(2, 6, 24))
    r.finish()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
engine.close()
