# Profile passes of a round (run on the GPU box through gpurun:  bash tools/scripts/profile_round.sh <dir under gpurun_out>).
# Round 5: ALL passes run the bench command itself (rounds 1-4 took the PMC passes from the torch-free
# probe) -- `bench.py --profile-legs` adds 5 jobs of the two-kernel path and 5 HBM-resident decodes behind the timed
# region, so every database holds k_encode_fused, k_quantize, k_cdf_encode and k_decode dispatches of the bench workload.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --profile-legs"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/stats.log 2>&1
P="$B --ramp-ms 0 --steps 5 --warmup 2"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $P > $O/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fe -o fe -- $P > $O/fe.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- $P > $O/wr.log 2>&1
cd $R; find gpurun_out/${1:-prof} -name "*.db" | xargs ls -la
