# Round validation on the GPU box (through gpurun):  bash tools/scripts/round_run.sh
# full GPU suite, smoke, the driver-style bench, then the profile passes behind profiles/ (render with
# tools/scripts/render_profiles.sh rprof r0N_x afterwards)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/round_pytest.log 2>&1; tail -2 gpurun_out/round_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3 4 5; do timeout 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['encode_paths']['two_kernels_ms'], d['encode_paths']['fused_ms'])"; done
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/round_bench.json 2> gpurun_out/round_bench.err
bash tools/scripts/profile_round.sh rprof > gpurun_out/round_prof.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/round_bench.json", "gpurun_out/rprof/stats.log"):
    txt = open(f).read()
    d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("encode_paths", {}).get("two_kernels_ms"), d.get("seeds", {}).get("median"))
PY
