"""bench.py's overlap legs alone (store_hidden: a non-blocking engine.store() of the 16 k context beside proxy decode
steps; ttft_proxy: the warm prefix from the pinned and the HBM tier) -- a minute instead of the whole bench.

    python tools/probes/store_completion.py
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
    kv = bench.make_kv(dev, 0, "rand")
    res = bench.overlap_legs(dev, kv, bench.L * 2 * bench.CTX * bench.H * bench.D * 2)
    sh, tp = res["store_hidden"], res["ttft_proxy"]
    print(json.dumps({"store_hidden": {k: v for k, v in sh.items() if k != "note"},
                      "ttft_pinned": {k: tp[k] for k in ("retrieve_plus_one_step_ms", "ratio", "layerwise_ms", "layerwise_ratio",
                                                         "layerwise_ms_by_layers_per_range")},
                      "ttft_hbm": {k: v for k, v in tp["hbm_tier"].items() if k != "note"}}, indent=1))


if __name__ == "__main__":
    main()
