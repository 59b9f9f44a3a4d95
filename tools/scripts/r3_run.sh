cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3Q_pytest.log 2>&1; tail -3 gpurun_out/r3Q_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3Q_bench.json 2> gpurun_out/r3Q_bench.err
LMC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3Q_bench_dist.json 2> gpurun_out/r3Q_bench_dist.err
bash tools/scripts/profile_round.sh r3prof > gpurun_out/r3Q_prof.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r3Q_bench.json","gpurun_out/r3Q_bench_dist.json","gpurun_out/r3prof/stats.log"):
    try:
        txt=open(f).read()
        d=json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("encode_paths",{}).get("two_kernels_ms"), d.get("seeds",{}).get("median"), str(d.get("exchange"))[:200])
    except Exception as e: print(f, "ERR", e)
PY
