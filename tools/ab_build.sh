#!/bin/bash
# A/B of one compile-time switch on one GPU box: tools/ab_build.sh MACRO V1 V2 ...  (the library is rebuilt for each value, then bench.py --steps 10, three rounds)
cd $GRAFT_REPO_ROOT
N=$1; shift
LIBP=regneuralde.jl_amd/lib/librnde.so
for v in "$@"; do
  RNDE_EXTRA_FLAGS="-D$N=$v" python -c "
import importlib.util
sp = importlib.util.spec_from_file_location('_b', 'regneuralde.jl_amd/build.py'); b = importlib.util.module_from_spec(sp); sp.loader.exec_module(b); b.build(force=True)" > /dev/null 2>&1
  cp $LIBP /tmp/librnde_$v.so
done
for rep in 1 2 3; do
  for v in "$@"; do
    cp /tmp/librnde_$v.so $LIBP
    timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$N=$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  fwd', round(d.get('us_per_attempt_fwd', 0), 2), 'rev', round(d.get('us_per_attempt_rev', 0), 2), 'nfe', d.get('mean_nfe'))"
  done
done
