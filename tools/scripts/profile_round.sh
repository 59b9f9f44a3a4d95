R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $B > $O/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fe -o fe -- $B > $O/fe.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- $B > $O/wr.log 2>&1
D="python $R/tools/probes/decode_rate.py --reps 3"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/dsq -o dsq -- $D > $O/dsq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/dfe -o dfe -- $D > $O/dfe.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/dwr -o dwr -- $D > $O/dwr.log 2>&1
cd $R; find gpurun_out/r2c -name "*.db" | xargs ls -la
