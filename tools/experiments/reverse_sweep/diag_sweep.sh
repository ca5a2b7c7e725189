# diagnostic build: librnde_diag.so = the product objects with the reverse-sweep unit rebuilt -DRNDE_DIAG_SWEEP (cycle stamps); run with
#   RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so RNDE_DIAG_SWEEP=1 python tools/diag_sweep.py
set -e
cd "$(dirname "$0")/.."
L=regneuralde.jl_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed -DRNDE_DIAG_SWEEP -c regneuralde.jl_amd/csrc/rnde_bstage_sweep.hip -o $L/obj/rnde_bstage_sweep_diag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_diag.so $L/obj/rnde.o $L/obj/rnde_stage_solve.o $L/obj/rnde_bstage_sweep_diag.o $L/obj/rnde_sde.o $L/obj/rnde_comm.o $L/obj/rnde_tapes.o -ldl
echo built $L/librnde_diag.so
