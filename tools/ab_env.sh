#!/bin/bash
# A/B of environment switches on the SAME GPU box: alternating runs of bench.py.  Usage: ab_env.sh "RNDE_X=1" "RNDE_Y=2" ...
# (RNDE_NOOP=1 stands for the default configuration)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "$@"; do
    env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('[$v]', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  attempt', round(d['roofline']['us_per_attempt'], 2), 'us  nfe', d['mean_nfe'], 'launches/attempt', d['roofline']['launches_per_unit'])"
  done
done
