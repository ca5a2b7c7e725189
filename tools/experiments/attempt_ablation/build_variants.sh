#!/bin/bash
# Ablation variants of the forward attempt kernel (no polls / no tanh / no tape / MFMAs alone).  The switches are NOT in the product sources:
# this script copies regneuralde.jl_amd/csrc to a scratch directory, applies rnde_stage_persist.h.patch there (it adds the RNDE_ABL_* switches;
# the variants' RESULTS are wrong by construction, only their time is read) and builds librnde_abl_NAME.so from the copy, next to the product
# library so that the variants travel with the gpurun snapshot.  Run HERE (CPU container), then on the GPU box:
#     python tools/experiments/attempt_ablation/ablate_attempt.py > profiles/r04_attempt_ablation.csv
set -e
cd "$(dirname "$0")/../../.."
HERE=tools/experiments/attempt_ablation
L=regneuralde.jl_amd/lib
T=$(mktemp -d /tmp/rnde_abl.XXXXXX)      # same shape as the repository: rnde.hip includes ../../include/rnde.h
S=$T/regneuralde.jl_amd/csrc
mkdir -p $S $T/include
cp regneuralde.jl_amd/csrc/* $S/
cp include/rnde.h $T/include/
patch -d $S -p0 < $HERE/rnde_stage_persist.h.patch
[ -f $L/obj/rnde_sde.o ] || python regneuralde.jl_amd/build.py --incremental
build() {
    N=$1; shift
    { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed "$@" -c $S/rnde.hip -o $L/obj/rnde_$N.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_$N.so $L/obj/rnde_$N.o $L/obj/rnde_reverse.o $L/obj/rnde_stage_solve.o $L/obj/rnde_latent.o $L/obj/rnde_sde.o $L/obj/rnde_comm.o $L/obj/rnde_tapes.o -ldl; } > /tmp/abl_$N.log 2>&1 || { echo "FAILED $N"; tail -5 /tmp/abl_$N.log; }
}
build abl_base &
build abl_nopoll -DRNDE_ABL_NOPOLL &
build abl_notanh -DRNDE_ABL_NOTANH &
build abl_notape -DRNDE_ABL_NOTAPE &
wait
build abl_nopoll_notanh -DRNDE_ABL_NOPOLL -DRNDE_ABL_NOTANH &
build abl_nopoll_notanh_notape -DRNDE_ABL_NOPOLL -DRNDE_ABL_NOTANH -DRNDE_ABL_NOTAPE &
build abl_mfmaonly -DRNDE_ABL_MFMAONLY -DRNDE_ABL_NOPOLL &
wait
rm -rf $T
ls -la $L/librnde_abl_*.so
