cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3T_pytest.log 2>&1; tail -3 gpurun_out/r3T_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
