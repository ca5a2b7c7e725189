cd $GRAFT_REPO_ROOT
for i in 1 2 3; do RNDE_X3=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys, json; o = json.loads(sys.stdin.read()); print('RNDE_X3=0', {k: round(o[k], 4) for k in ('value', 'ms_per_step', 'mean_nfe', 'us_per_attempt_fwd', 'us_per_attempt_rev', 'rev_rest_ms')}, o.get('solve_diag'))"; done
