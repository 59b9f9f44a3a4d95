cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ctx in 16384 16640 16128; do
AB="tools/probes/encode_ab 32 8 128 $ctx 256 0 20 0 2"
for v in main rot7 rot13 main rot7; do
  if [ $v = main ]; then L=""; else L="$PWD/build_alt/$v"; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout 120 $AB > gpurun_out/r3C_$v.log 2>&1; echo "$ctx $v: $(grep -E '^fused|^two' gpurun_out/r3C_$v.log | awk '{print $1, $2, $6, $9, $10}' | tr '\n' ' ')"; done; done
