# Profile passes of a round (run on the GPU box through gpurun:  bash tools/scripts/profile_round.sh <dir under gpurun_out>): kernel stats of the default bench command, and the
# three PMC passes over the torch-free A/B probe (tools/probes/encode_ab: the bench workload through BOTH launch
# paths of lmc_encode_chunks, so one database holds k_encode_fused next to k_quantize + k_cdf_encode).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
AB="$R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $AB > $O/sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fe -o fe -- $AB > $O/fe.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- $AB > $O/wr.log 2>&1
cd $R; find gpurun_out/${1:-prof} -name "*.db" | xargs ls -la
