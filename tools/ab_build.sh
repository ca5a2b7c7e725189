#!/bin/bash
# A/B of one compile-time switch on one GPU box: tools/ab_build.sh MACRO V1 V2 ...
# Each value is built into ITS OWN library and object directory (/tmp/rnde_ab/MACRO_V/librnde.so via RNDE_LIB; the default
# regneuralde.jl_amd/lib/librnde.so and its objects are never touched), then bench.py --steps 10 alternates between them, three rounds.
# (Cheaper: build the variants in the CPU container with tools/build_variant.sh -- they travel with the snapshot -- and pass
#  AB_PREBUILT=1 with the variant NAMES instead of values: tools/ab_build.sh - nameA nameB.)
cd $GRAFT_REPO_ROOT
N=$1; shift
libof() { if [ -n "$AB_PREBUILT" ]; then echo "$GRAFT_REPO_ROOT/regneuralde.jl_amd/lib/librnde_$1.so"; else echo "/tmp/rnde_ab/${N}_$1/librnde.so"; fi; }
if [ -z "$AB_PREBUILT" ]; then
  for v in "$@"; do
    mkdir -p /tmp/rnde_ab/${N}_$v
    RNDE_LIB=$(libof $v) RNDE_EXTRA_FLAGS="-D$N=$v" python -c "
import importlib.util
sp = importlib.util.spec_from_file_location('_b', 'regneuralde.jl_amd/build.py'); b = importlib.util.module_from_spec(sp); sp.loader.exec_module(b); b.build(force=True)" > /dev/null 2>&1
  done
fi
for rep in 1 2 3; do
  for v in "$@"; do
    RNDE_LIB=$(libof $v) timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$N=$v', round(d['value']), 'samples/s', round(d['ms_per_step'], 3), 'ms  fwd', round(d.get('us_per_attempt_fwd', 0), 2), 'rev', round(d.get('us_per_attempt_rev', 0), 2), 'nfe', d.get('mean_nfe'))"
  done
done
