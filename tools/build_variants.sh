#!/bin/bash
# Alternative builds of liblmc_hip.so for A/B timing on the GPU box (tools/probes/encode_ab picks one up through
# LD_LIBRARY_PATH; the Python side through LMCACHE_AMD_SO):  tools/build_variants.sh name "-DFLAG=1 ..." [name flags ...]
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -fPIC -shared -Wall -Wno-unused-function -Wl,-rpath,/opt/rocm/lib"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  mkdir -p build_alt/$name
  ( /opt/rocm/bin/hipcc $FLAGS $defs lmcache_amd/csrc/lmc_api.hip -o build_alt/$name/liblmc_hip.so 2>&1 | grep -v hip-link; echo "built $name ($defs)" ) &
done
wait
