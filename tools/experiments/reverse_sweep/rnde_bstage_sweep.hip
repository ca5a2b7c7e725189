// rnde_bstage_sweep.hip -- translation unit of the stage engine's one-launch reverse sweep (rnde_bstage_sweep.h); rnde.hip calls the launcher.
#include <hip/hip_runtime.h>
#include "../../../include/rnde.h"
// (own namespace name for this unit's copy of the non-template kernels in the shared headers, parameter blocks by address: see rnde_stage_solve.hip)
#define rnde rnde_sweep_tu
#include "rnde_bstage_sweep.h"

using namespace rnde;

// reverses the attempts n_hi, n_hi - 1, .. n_lo; grid: 8 * 7 * ceil(C / 8) workgroups of 7 waves, all resident at once
extern "C" hipError_t rnde_launch_bstage_sweep(const void* bstage_params, int n_hi, int n_lo, const void* sweep_args_dev, const void* persist_sync,
                                               const void* solve_sync, int act2, hipStream_t s) {
    const BStageParams& Q = *(const BStageParams*)bstage_params;
    const PersistSync& Y = *(const PersistSync*)persist_sync;
    const SolveSync& Z = *(const SolveSync*)solve_sync;
    const size_t lds = sizeof(float) * (2 * kSCB * (16 * 7 + 4) + 64 + 17 * 7 * 256);      // operand images, scratch, 17 tape slots of 1 KiB per wave
    static const hipError_t attr = [] {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_bstage_sweep_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e == hipSuccess ? hipFuncSetAttribute((const void*)rnde_bstage_sweep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) : e;
    }();
    if (attr != hipSuccess) return attr;
    const dim3 grid(8 * 7 * ((Q.C + 7) / 8));
    if (act2) hipLaunchKernelGGL((rnde_bstage_sweep_kernel<1>), grid, dim3(64 * 7), lds, s, Q, n_hi, n_lo, (const SweepArgs*)sweep_args_dev, Y, Z);
    else hipLaunchKernelGGL((rnde_bstage_sweep_kernel<0>), grid, dim3(64 * 7), lds, s, Q, n_hi, n_lo, (const SweepArgs*)sweep_args_dev, Y, Z);
    return hipGetLastError();
}
