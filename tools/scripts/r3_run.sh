cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do timeout 200 tools/probes/encode_modes 4 > gpurun_out/r3J_modes$i.log 2>&1; cat gpurun_out/r3J_modes$i.log; echo ---; done
