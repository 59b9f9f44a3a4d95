cd $GRAFT_REPO_ROOT; R=$PWD; mkdir -p gpurun_out/r3U
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 2"
for v in main head main head main head; do
  if [ $v = main ]; then L=""; else L="$PWD/build_alt/$v"; fi
  LD_LIBRARY_PATH=$L:$LD_LIBRARY_PATH timeout 120 $AB > gpurun_out/r3U_$v.log 2>&1; echo "$v: $(grep -E '^fused|^two|PARITY' gpurun_out/r3U_$v.log | awk '{print $1, $2, $9, $10}' | tr '\n' ' ')"; done
timeout 120 tools/probes/encode_ab 32 8 128 16384 256 1 5 2 > gpurun_out/r3U_special.log 2>&1; grep -E "fused|PARITY" gpurun_out/r3U_special.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3U_pytest.log 2>&1; tail -3 gpurun_out/r3U_pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/r3U/sq -o sq -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $R/gpurun_out/r3U/sq.log 2>&1
cd $R; python tools/rocpd_stats.py gpurun_out/r3U/sq/sq_results.db --min-grid 2000000 --per 16777216 | grep "k_encode_fused.*SQ_INSTS"
