cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_c_abi_store_load.py tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/r3N_pytest.log 2>&1; tail -15 gpurun_out/r3N_pytest.log
