cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AB="tools/probes/encode_ab 32 8 128 16384 256 0 20 0 3"
for c in 3 1 3 1 3 1 3 1; do LMC_WS_CANDIDATES=$c timeout 120 $AB > gpurun_out/r3Z_c$c.log 2>&1; echo "cands $c: $(grep -E '^fused|PARITY' gpurun_out/r3Z_c$c.log | awk '{print $1, $2}' | tr '\n' ' ')"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_c_abi_store_load.py tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r3Z_pytest.log 2>&1; tail -3 gpurun_out/r3Z_pytest.log
