# A/B variant of librnde.so: tools/build_variant.sh NAME [-DFOO=1 ...]  ->  regneuralde.jl_amd/lib/librnde_NAME.so (only rnde.hip -- the forward side -- is recompiled)
set -e
cd "$(dirname "$0")/.."
L=regneuralde.jl_amd/lib
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed "$@" -c regneuralde.jl_amd/csrc/rnde.hip -o $L/obj/rnde_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/librnde_$N.so $L/obj/rnde_$N.o $L/obj/rnde_reverse.o $L/obj/rnde_stage_solve.o $L/obj/rnde_latent.o $L/obj/rnde_sde.o $L/obj/rnde_comm.o $L/obj/rnde_tapes.o -ldl
echo built $L/librnde_$N.so
