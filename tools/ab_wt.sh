#!/bin/bash
cd $GRAFT_REPO_ROOT
for wt in 7 4 5 6 8; do
  RNDE_PERSIST=0 RNDE_STAGE_WT=$wt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('multi-launch WT=$wt', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  attempt', round(d['roofline']['us_per_attempt'], 2), 'us  nfe', d['mean_nfe'])"
done
