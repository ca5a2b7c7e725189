cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_x3.py tests/test_gpu_forward.py tests/test_gpu_solve.py tests/test_gpu_coupled.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 600 python -m pytest tests -x -q -m gpu -k "two_tile or full_size or large_batch" 2>&1 | grep -E "passed|failed"
for v in "" "RNDE_NO_EPART_REDUCE=1" "" "RNDE_NO_EPART_REDUCE=1"; do
env $v timeout 300 python bench.py --batch 4096 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys, json; o = json.loads(sys.stdin.read()); r = o['roofline']; print('$v', {k: round(o[k], 4) for k in ('value', 'ms_per_step', 'mean_nfe', 'us_per_attempt_fwd', 'us_per_attempt_rev', 'rev_rest_ms')}, round(r['frac'], 4))"
done
