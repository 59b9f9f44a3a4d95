cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { env $2 timeout 200 tools/probes/encode_modes 4 > gpurun_out/r3K_$1.log 2>&1; echo "== $1 ($2): $(grep -o 'fused [0-9.]*' gpurun_out/r3K_$1.log | cut -d' ' -f2 | tr '\n' ' ')"; }
run a "X=1"
run b "LMC_SYM_PAD=0 LMC_SCRATCH_PAD=0"
run c "X=1"
run d "LMC_SYM_PAD=0 LMC_SCRATCH_PAD=0"
run e "LMC_SCRATCH_PAD=0"
run f "LMC_SYM_PAD=0"
run g "LMC_SYM_PAD=65792 LMC_SCRATCH_PAD=1024"
run h "LMC_SYM_PAD=20736 LMC_SCRATCH_PAD=4352"
tail -3 gpurun_out/r3K_a.log
