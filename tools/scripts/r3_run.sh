cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_c_abi_store_load.py -m gpu -x -q > gpurun_out/r3O_pytest.log 2>&1; tail -6 gpurun_out/r3O_pytest.log
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "config0 or config3" > gpurun_out/r3O_pytest2.log 2>&1; tail -3 gpurun_out/r3O_pytest2.log
timeout 120 tools/probes/encode_ab 32 32 128 4096 256 1 10 0 > gpurun_out/r3O_c4096.log 2>&1; grep -E "two-kernel|fused|decode|PARITY" gpurun_out/r3O_c4096.log
timeout 120 tools/probes/encode_ab 16 16 128 8192 256 0 10 2 > gpurun_out/r3O_c2048.log 2>&1; grep -E "two-kernel|fused|decode|PARITY" gpurun_out/r3O_c2048.log
