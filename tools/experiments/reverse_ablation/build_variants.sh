#!/bin/bash
# Ablation variants of the REVERSE attempt kernel rnde_bstage_attempt_kernel<1,1> (no polls / no tape DMA / no tape stores / no scalar chain / MFMAs
# alone / START alone / no END).  The switches are NOT in the product sources: this script copies regneuralde.jl_amd/csrc to a scratch directory,
# applies rnde_bstage_persist.h.patch (RNDE_RABL_*) and the forward ablation's rnde_stage_persist.h.patch (RNDE_ABL_NOPOLL lives in slab_poll_sum,
# which both directions share) there and builds librnde_rabl_NAME.so from the copy (only rnde_reverse.hip is recompiled; the forward side is the
# product's object), next to the product library so that the variants travel with the gpurun snapshot.  The variants' RESULTS are wrong by
# construction, only their time is read.  Run HERE (CPU container), then on the GPU box:
#     python tools/experiments/reverse_ablation/ablate_reverse.py > profiles/r05_rev_attempt_ablation.csv
set -e
cd "$(dirname "$0")/../../.."
HERE=tools/experiments/reverse_ablation
L=regneuralde.jl_amd/lib
T=$(mktemp -d /tmp/rnde_rabl.XXXXXX)      # same shape as the repository: the sources include ../../include/rnde.h
S=$T/regneuralde.jl_amd/csrc
mkdir -p $S $T/include
cp regneuralde.jl_amd/csrc/* $S/
cp include/rnde.h $T/include/
patch -s -d $S -p0 < $HERE/rnde_bstage_persist.h.patch
patch -s -d $S -p0 < tools/experiments/attempt_ablation/rnde_stage_persist.h.patch
[ -f $L/obj/rnde_sde.o ] || python regneuralde.jl_amd/build.py --incremental
build() {
    N=$1; shift
    { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed "$@" -c $S/rnde_reverse.hip -o $L/obj/rnde_reverse_$N.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_$N.so $L/obj/rnde.o $L/obj/rnde_reverse_$N.o $L/obj/rnde_stage_solve.o $L/obj/rnde_latent.o $L/obj/rnde_sde.o $L/obj/rnde_comm.o $L/obj/rnde_tapes.o -ldl; } > /tmp/rabl_$N.log 2>&1 || { echo "FAILED $N"; tail -5 /tmp/rabl_$N.log; }
}
build rabl_base &
build rabl_nopoll -DRNDE_ABL_NOPOLL &
build rabl_nodma -DRNDE_RABL_NODMA &
build rabl_nostore -DRNDE_RABL_NOSTORE &
wait
build rabl_noscalar -DRNDE_RABL_NOSCALAR &
build rabl_nopoll_nodma_nostore -DRNDE_ABL_NOPOLL -DRNDE_RABL_NODMA -DRNDE_RABL_NOSTORE &
build rabl_mfmaonly -DRNDE_RABL_MFMAONLY -DRNDE_ABL_NOPOLL &
build rabl_startonly -DRNDE_RABL_STARTONLY &
wait
build rabl_noend -DRNDE_RABL_NOEND &
build rabl_startonly_noscalar -DRNDE_RABL_STARTONLY -DRNDE_RABL_NOSCALAR &
wait
rm -rf $T
ls -la $L/librnde_rabl_*.so
