cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 3"
for s in 50 0 30 70 50 0 30 70; do LMC_FUSED_STAGGER_US=$s timeout 120 $AB > gpurun_out/r3ab_s$s.log 2>&1; echo "stagger $s: $(grep -E '^fused' gpurun_out/r3ab_s$s.log | awk '{print $2}' | tr '\n' ' ')"; done
