# Quick kernel A/B on the GPU box (through gpurun):  bash tools/scripts/quick_ab.sh <tag>
# parity of both launch paths through the torch-free probe, then one SQ counter pass of the same probe.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
$R/tools/probes/encode_ab 32 8 128 16384 256 0 40 0 3 2>&1 | tail -12
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $O/sq.log 2>&1
cd $R; python tools/rocpd_stats.py $O/sq/sq_results.db --min-grid 2000000 --per 16777216 | grep "INSTS_VALU\|WAVE_CYCLES\|WAIT_ANY"
