# diagnostic build of librnde.so with clock64 stamps (RNDE_DIAG): regneuralde.jl_amd/lib/librnde_diag.so
set -e
cd "$(dirname "$0")/.."
L=regneuralde.jl_amd/lib
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed -DRNDE_DIAG"
/opt/rocm/bin/hipcc $F -c regneuralde.jl_amd/csrc/rnde.hip -o $L/obj/rnde_diag.o &
/opt/rocm/bin/hipcc $F -c regneuralde.jl_amd/csrc/rnde_stage_solve.hip -o $L/obj/rnde_stage_solve_diag.o &
/opt/rocm/bin/hipcc $F -c regneuralde.jl_amd/csrc/rnde_reverse.hip -o $L/obj/rnde_reverse_diag.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_diag.so $L/obj/rnde_diag.o $L/obj/rnde_reverse_diag.o $L/obj/rnde_stage_solve_diag.o $L/obj/rnde_latent.o $L/obj/rnde_sde.o $L/obj/rnde_comm.o $L/obj/rnde_tapes.o -ldl
echo built $L/librnde_diag.so
