# Instruction-fetch counters of the encode / decode kernels (through gpurun): bash tools/scripts/icache_pmc.sh <tag> [lib-dir]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ic}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export LD_LIBRARY_PATH=$R/$2:$LD_LIBRARY_PATH   # a directory under the repo, e.g. build_alt/base
timeout 120 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/ic -o ic -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $O/ic.log 2>&1
cd $R; python tools/rocpd_stats.py $O/ic/ic_results.db --min-grid 2000000 --per 16777216 | grep -v "^| .fill\|^$" | tail -40
