"""The CPU baseline's thread-scaling table alone (bench.py: cpu_baseline), e.g. to see what OMP_PROC_BIND / OMP_PLACES do:

    OMP_PROC_BIND=spread OMP_PLACES=cores python tools/probes/cpu_baseline_scaling.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.cpu_baseline(4 * len(os.sched_getaffinity(0)))
    out.pop("reference_formula", None)
    out.pop("torch_serde", None)
    print(json.dumps(out, indent=1))
