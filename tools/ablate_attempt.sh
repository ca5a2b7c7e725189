#!/bin/bash
# Builds the ablation variants of the forward attempt kernel HERE (CPU container; they travel with the gpurun snapshot as
# regneuralde.jl_amd/lib/librnde_abl_*.so) -- then on the GPU box: python tools/ablate_attempt.py > profiles/r03_attempt_ablation.csv
cd "$(dirname "$0")/.."
build() { bash tools/build_variant.sh "$@" > /tmp/abl_$1.log 2>&1 || { echo "FAILED $1"; tail -5 /tmp/abl_$1.log; }; }
build abl_base &
build abl_nopoll -DRNDE_ABL_NOPOLL &
build abl_notanh -DRNDE_ABL_NOTANH &
build abl_notape -DRNDE_ABL_NOTAPE &
wait
build abl_nopoll_notanh -DRNDE_ABL_NOPOLL -DRNDE_ABL_NOTANH &
build abl_nopoll_notanh_notape -DRNDE_ABL_NOPOLL -DRNDE_ABL_NOTANH -DRNDE_ABL_NOTAPE &
build abl_mfmaonly -DRNDE_ABL_MFMAONLY -DRNDE_ABL_NOPOLL &
wait
ls -la regneuralde.jl_amd/lib/librnde_abl_*.so
