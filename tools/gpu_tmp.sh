cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_x3.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys, json; o = json.loads(sys.stdin.read()); r = o['roofline']; print({k: round(o[k], 4) for k in ('value', 'ms_per_step', 'mean_nfe', 'us_per_attempt_fwd', 'us_per_attempt_rev', 'rev_rest_ms')}, round(r['frac'], 4), round(r['us_per_attempt_back_to_back'], 3))"
done
