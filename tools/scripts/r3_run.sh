cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3x_pytest.log 2>&1; tail -3 gpurun_out/r3x_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3x_bench.json 2> gpurun_out/r3x_bench.err
python - <<'PY'
import json
txt=open("gpurun_out/r3x_bench.json").read()
d=json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"]); t=d["ttft_proxy"]; print({k:t[k] for k in t if k.startswith(("layerwise","retrieve","one_step","warm","cold","pcie"))}); print(d["offload_c_abi"]); print(d["decode"]); print(d["store_hidden"])
PY
