#!/bin/bash
# a long training run of the headline workload with the library built for each value of a compile-time macro: tools/long_run.sh MACRO V1 V2 ... (STEPS=1500)
# (each value in its own library and object directory under /tmp/rnde_ab, selected with RNDE_LIB; the default library is never touched)
cd $GRAFT_REPO_ROOT
N=$1; shift
for v in "$@"; do
  mkdir -p /tmp/rnde_ab/${N}_$v
  RNDE_LIB=/tmp/rnde_ab/${N}_$v/librnde.so RNDE_EXTRA_FLAGS="-D$N=$v" python -c "
import importlib.util
sp = importlib.util.spec_from_file_location('_b', 'regneuralde.jl_amd/build.py'); b = importlib.util.module_from_spec(sp); sp.loader.exec_module(b); b.build(force=True)" > /dev/null 2>&1
  echo "$N=$v"; RNDE_LIB=/tmp/rnde_ab/${N}_$v/librnde.so python bench.py --steps ${STEPS:-1500} --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-420
done
