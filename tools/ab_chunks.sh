#!/bin/bash
# weight-gradient tail launch: number of K chunks (2 workgroups each) -- RNDE_WGRAD3_CHUNKS -- on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 32 64 96 128 192 256; do
    RNDE_WGRAD3_CHUNKS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('chunks=$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  fwd', round(d['us_per_attempt_fwd'], 2), 'rev', round(d['us_per_attempt_rev'], 2), 'rest', round(d['rev_rest_ms'], 3), 'nfe', d['mean_nfe'])"
  done
done
