cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do
timeout 300 python bench.py --batch 4096 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys, json; o = json.loads(sys.stdin.read()); r = o['roofline']; print({k: round(o[k], 4) for k in ('value', 'ms_per_step', 'mean_nfe', 'us_per_attempt_fwd', 'us_per_attempt_rev', 'rev_rest_ms')}, round(r['frac'], 4))"
done
