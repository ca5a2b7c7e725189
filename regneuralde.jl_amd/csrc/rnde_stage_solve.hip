// rnde_stage_solve.hip -- translation unit of the stage engine's one-launch solve (rnde_stage_solve.h); rnde.hip calls the launcher.
#include <hip/hip_runtime.h>
// The kernel headers define (non-template) kernels in namespace rnde; rnde.hip includes them too.  This translation unit gets its own
// copy under another namespace name (as rnde_sde.hip does), and the launcher takes the parameter blocks by address: both units
// compile the same struct definitions.
#define rnde rnde_solve_tu
#include "rnde_stage_solve.h"

using namespace rnde;

// grid: 8 * 7 * ceil(C / 8) workgroups of 7 waves, all resident at once (the host only calls this with at most 256 of them)
// x3: the Dense-layer products on the matrix cores (rnde_x3.h); Z.x3B / Z.x3D must then hold the split weights (rnde_launch_x3_pack)
extern "C" hipError_t rnde_launch_stage_solve(const void* stage_params, const void* persist_sync, const void* solve_sync, int act2, int x3, hipStream_t s) {
    const StageParams& Q = *(const StageParams*)stage_params;
    const PersistSync& Y = *(const PersistSync*)persist_sync;
    const SolveSync& Z = *(const SolveSync*)solve_sync;
    const size_t img = x3 ? (size_t)2 * kX3ImageFloats : (size_t)2 * kSCB * (16 * 7 + 4);
    const size_t lds = sizeof(float) * (img + 32) + 4 * sizeof(double) + 16 + 12 * sizeof(int);      // HL, GL, RED | SUMS | QP (+ pad) | SS
    const dim3 grid(8 * 7 * ((Q.C + 7) / 8));
    if (x3) {
        if (act2) hipLaunchKernelGGL((rnde_stage_solve_kernel<1, 1>), grid, dim3(64 * 7), lds, s, Q, Y, Z);
        else hipLaunchKernelGGL((rnde_stage_solve_kernel<0, 1>), grid, dim3(64 * 7), lds, s, Q, Y, Z);
    } else {
        if (act2) hipLaunchKernelGGL((rnde_stage_solve_kernel<1, 0>), grid, dim3(64 * 7), lds, s, Q, Y, Z);
        else hipLaunchKernelGGL((rnde_stage_solve_kernel<0, 0>), grid, dim3(64 * 7), lds, s, Q, Y, Z);
    }
    return hipGetLastError();
}
// the weights split into three bf16 planes (rnde_x3.h): one launch in front of a forward solve whose kernels run with x3; x3Bt / x3Dt (may be null):
// the transposed images of the reverse attempt kernel
extern "C" hipError_t rnde_launch_x3_pack(const float* p, void* x3B, void* x3D, void* x3Bt, void* x3Dt, int D, int H, int MT, int WT, int R, int HT, hipStream_t s) {
    const long long total = (long long)(MT + HT * R) * 4 * 64 * ((x3Bt && x3Dt) ? 2 : 1);
    X3PackDst dst{{(x3u4*)x3B, (x3u4*)x3D, (x3Bt && x3Dt) ? (x3u4*)x3Bt : nullptr, (x3Bt && x3Dt) ? (x3u4*)x3Dt : nullptr}};
    hipLaunchKernelGGL(rnde_x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, dst, D, H, MT, WT, R, HT);
    return hipGetLastError();
}
