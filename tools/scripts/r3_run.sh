cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_c_abi_store_load.py -m gpu -x -q > gpurun_out/r3P_pytest.log 2>&1; tail -4 gpurun_out/r3P_pytest.log
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r3P_pytest2.log 2>&1; tail -3 gpurun_out/r3P_pytest2.log
timeout 120 tools/probes/encode_ab 32 32 128 4096 256 1 10 0 2 > gpurun_out/r3P_c4096.log 2>&1; grep -E "two-kernel|PARITY" gpurun_out/r3P_c4096.log
timeout 120 tools/probes/encode_ab 80 1 128 32768 256 0 10 0 2 > gpurun_out/r3P_c128.log 2>&1; grep -E "two-kernel|PARITY" gpurun_out/r3P_c128.log
timeout 120 tools/probes/encode_ab 32 8 128 16384 256 0 10 0 2 > gpurun_out/r3P_main.log 2>&1; grep -E "two-kernel|fused|PARITY" gpurun_out/r3P_main.log
