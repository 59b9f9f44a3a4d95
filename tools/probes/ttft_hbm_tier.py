"""The HBM-tier warm-prefix TTFT proxy of bench.py (ttft_hbm_tier) alone: one JSON line, ~20 s.

    python tools/probes/ttft_hbm_tier.py
    # the timeline behind profiles/r05_ttft_timeline.md:
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d out -o t -- python $REPO/tools/probes/ttft_hbm_tier.py
    python tools/ttft_timeline.py out/t_kernel_trace.csv
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from lmcache_amd.config import LMCacheEngineMetadata
    kv = bench.make_kv(dev, 0, "rand")
    meta = LMCacheEngineMetadata(bench.MODEL, 1, 0, "vllm", "bfloat16")
    proxy = bench.DecodeStepProxy(dev)
    proxy.time_steps(3)
    alone = bench.median(proxy.time_steps(10))
    r = bench.ttft_hbm_tier(dev, kv, proxy, alone, meta)
    print(json.dumps({"one_step_ms": round(alone, 3),
                      "retrieve_then_step_ms": r["retrieve_then_step_ms"], "layerwise_ms": r["layerwise_ms"],
                      "layerwise_ratio": r["layerwise_ratio"], "best": r["layers_per_launch"],
                      "by_schedule": r["layerwise_ms_by_layers_per_launch"],
                      "host_ms": r["host_ms_before_the_model_can_start"]}))


if __name__ == "__main__":
    main()
