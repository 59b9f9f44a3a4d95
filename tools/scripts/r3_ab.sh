cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 3"
for v in 0 1 0 1 0 1; do LMC_FUSED_SLOTS=$v timeout 120 $AB > gpurun_out/r3ab_slots$v.log 2>&1; echo "slots $v: $(grep -E '^fused|PARITY' gpurun_out/r3ab_slots$v.log | awk '{print $1,$2}' | tr '\n' ' ')"; done
LMC_FUSED_SLOTS=1 timeout 120 tools/probes/encode_ab 32 8 128 16384 256 1 5 2 2>&1 | grep -E "PARITY|status"
