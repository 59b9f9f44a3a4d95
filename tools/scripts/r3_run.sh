cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['encode_paths']['two_kernels_ms'], d['encode_paths']['fused_ms'])"; done
