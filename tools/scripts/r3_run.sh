cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 3"
for s in 0 20 30 40 60 0 30; do LMC_FUSED_STAGGER_US=$s timeout 120 $AB > gpurun_out/r3s_st$s.log 2>&1; echo "stagger $s: $(grep -E '^fused' gpurun_out/r3s_st$s.log | awk '{print $2}' | tr '\n' ' ')"; done
for v in serial main; do
  if [ $v = main ]; then L=""; else L="$PWD/build_alt/$v"; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout 120 $AB > gpurun_out/r3s_$v.log 2>&1; echo "$v: $(grep -E '^fused|^two' gpurun_out/r3s_$v.log | awk '{print $1, $2}' | tr '\n' ' ')"; done
