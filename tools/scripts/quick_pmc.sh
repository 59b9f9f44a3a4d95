# One SQ counter pass over the torch-free probe for a build (through gpurun):  bash tools/scripts/quick_pmc.sh <tag> [lib-dir]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-qp}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export LD_LIBRARY_PATH=$R/$2:$LD_LIBRARY_PATH   # a directory under the repo, e.g. build_alt/base
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $O/sq.log 2>&1
cd $R; python tools/rocpd_stats.py $O/sq/sq_results.db --min-grid 2000000 --per 16777216 | grep "fused.*INSTS\|fused.*WAVE_CYCLES\|fused.*WAIT_ANY "
