#!/bin/bash
# A/B builds of librnde.so on the SAME GPU box (boxes differ by a few percent): alternating runs of bench.py
# usage (on the GPU box): tools/ab_bench.sh LIB_A LIB_B ...      extra environment for all runs via ENVX="K=V ..."
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "$@"; do
    env $ENVX RNDE_LIB=$PWD/$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  fwd', round(d['us_per_attempt_fwd'], 2), 'rev', round(d['us_per_attempt_rev'], 2), 'rest', round(d['rev_rest_ms'], 3), 'nfe', d['mean_nfe'])"
  done
done
