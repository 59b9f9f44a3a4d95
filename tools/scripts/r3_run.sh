cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_c_abi_store_load.py -m gpu -x -q > gpurun_out/r3A_pytest.log 2>&1; tail -3 gpurun_out/r3A_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3A_bench.json 2> gpurun_out/r3A_bench.err
python - <<'PY'
import json
txt=open("gpurun_out/r3A_bench.json").read()
d=json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"]); t=d["ttft_proxy"]; print({k:t[k] for k in t if k.startswith(("layerwise","retrieve","one_step","warm","pcie"))}); print(d.get("offload_pack")); print(d["store_hidden"])
PY
tail -3 gpurun_out/r3A_bench.err
