#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
 for cfg in "A 0" "B 768" "B 1536" "B 1100" "B 1792"; do
  set -- $cfg
  RNDE_WGRAD_WGS=$2 RNDE_LIB=$PWD/tools/micro/librnde_$1.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1 wgs=$2', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms')"
 done
done
