#!/bin/bash
# A/B builds of librnde.so on the SAME GPU box (boxes differ by a few percent): alternating runs of bench.py
# VARIANTS="A B" name tools/micro/librnde_<v>.so; extra environment for all runs via ENVX="K=V ..."
cd $GRAFT_REPO_ROOT
VARIANTS=${VARIANTS:-"A B"}
for rep in 1 2 3; do
  for v in $VARIANTS; do
    env $ENVX RNDE_LIB=$PWD/tools/micro/librnde_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  attempt', round(d['roofline']['us_per_attempt'], 2), 'us  nfe', d['mean_nfe'])"
  done
done
