cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 3"
for v in main qnt dnt main qnt dnt main qnt dnt; do
  if [ $v = main ]; then L=""; else L="$PWD/build_alt/$v"; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout 120 $AB > gpurun_out/r3ab_$v.log 2>&1; echo "$v: $(grep -E '^fused|^two|^decode|PARITY' gpurun_out/r3ab_$v.log | awk '{print $1, $2}' | tr '\n' ' ')"; done
