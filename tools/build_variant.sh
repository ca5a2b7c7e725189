#!/bin/bash
# a variant of librnde.so with extra compile flags for ONE translation unit: tools/build_variant.sh NAME rnde_reverse.hip -DFOO=1 ...
# -> regneuralde.jl_amd/lib/librnde_NAME.so (travels with the gpurun snapshot; select it with RNDE_LIB or tools/ab_build.sh with AB_PREBUILT=1)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
L=regneuralde.jl_amd/lib
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed"
/opt/rocm/bin/hipcc $F "$@" -c regneuralde.jl_amd/csrc/$SRC -o $L/obj/${SRC%.hip}_$NAME.o
OBJS=""
for o in rnde rnde_reverse rnde_stage_solve rnde_latent rnde_sde rnde_comm rnde_tapes; do
  if [ "$o.hip" == "$SRC" ]; then OBJS="$OBJS $L/obj/${o}_$NAME.o"; else OBJS="$OBJS $L/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_$NAME.so $OBJS -ldl
echo built $L/librnde_$NAME.so
