#!/bin/bash
# A/B of one environment switch on one GPU box: tools/ab_env.sh NAME V1 V2 ...   (alternating runs of bench.py --steps 10)
cd $GRAFT_REPO_ROOT
N=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $N=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$N=$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  fwd', round(d['us_per_attempt_fwd'], 2), 'rev', round(d['us_per_attempt_rev'], 2), 'rest', round(d['rev_rest_ms'], 3), 'nfe', d['mean_nfe'])"
  done
done
