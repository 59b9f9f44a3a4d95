cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 2"
for v in main head persist main head persist main persist; do
  if [ $v = main ]; then L=""; else L="$PWD/build_alt/$v"; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout 120 $AB > gpurun_out/r3I_$v.log 2>&1; echo "$v: $(grep -E '^fused|^two|PARITY' gpurun_out/r3I_$v.log | awk '{print $1, $2, $9, $10}' | tr '\n' ' ')"; done
