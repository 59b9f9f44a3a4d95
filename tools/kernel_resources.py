#!/usr/bin/env python
"""Registers / scratch / LDS / occupancy of every kernel in a hipcc -S dump (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing --cuda-device-only -S \
          -o /tmp/lmc.s lmcache_amd/csrc/lmc_api.hip
    python tools/kernel_resources.py /tmp/lmc.s [name-substring]
"""
import re
import sys

path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
row = {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, row = m.group(1), {}
        continue
    m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)\s*[:=] (\d+)", line)
    if m and name:
        row[m.group(1)] = int(m.group(2))
        if m.group(1) == "Occupancy" and key in name:
            print(f"{name[:70]:70s} vgpr {row.get('NumVgprs', -1):3d} sgpr {row.get('NumSgprs', -1):3d} scratch "
                  f"{row.get('ScratchSize', -1):4d} occ {row.get('Occupancy')} lds {row.get('LDSByteSize', -1):6d} "
                  f"code {row.get('codeLenInByte')}")
