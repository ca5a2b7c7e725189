// checker for slab_put / slab_poll_sum exactly as the persistent kernels use them
#include "../../regneuralde.jl_amd/csrc/rnde_stage_persist.h"
#include <cstdio>
using namespace rnde;
constexpr int R = 7, C = 32, HT = 7;
__device__ __forceinline__ float val(int it, int rb, int w, int lane, int q) { return (float)((it % 50) * 4 + rb) + 0.5f * w + 16.f * q + 64.f * (lane & 7); }
__global__ __launch_bounds__(448) void k(PersistSync Y, unsigned* errs, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = (blockIdx.x >> 3) % R, ct = 8 * ((blockIdx.x >> 3) / R) + (blockIdx.x & 7);
    unsigned nerr = 0;
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        const unsigned tag = Y.seq_base + it + 1;
        const size_t tile0 = (((size_t)par * C + ct) * R + rb) * HT;
        slab_put(Y.tslab, tile0 + w, lane, (f32x4){val(it, rb, w, lane, 0), val(it, rb, w, lane, 1), val(it, rb, w, lane, 2), val(it, rb, w, lane, 3)}, tag);
        f32x4 zs;
        if (!slab_poll_sum(Y, par, C, R, HT, ct, w, lane, tag, zs)) break;
        f32x4 ref = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) ref += (f32x4){val(it, r, w, lane, 0), val(it, r, w, lane, 1), val(it, r, w, lane, 2), val(it, r, w, lane, 3)};
        for (int q = 0; q < 4; ++q) if (zs[q] != ref[q]) ++nerr;
        __syncthreads();
    }
    if (nerr) atomicAdd(errs, nerr);
}
int main() {
    float* tslab; unsigned *abortf, *errs, *xcc;
    const int G = R * C;
    const size_t bytes = (size_t)2 * C * R * HT * 128 * 16;
    hipMalloc(&tslab, bytes); hipMalloc(&abortf, 8); hipMalloc(&errs, 4); hipMalloc(&xcc, G * 4);
    hipMemset(tslab, 0, bytes); hipMemset(abortf, 0, 8); hipMemset(errs, 0, 4);
    PersistSync Y{tslab, abortf, xcc, 0};
    hipLaunchKernelGGL(k, dim3(G), dim3(448), 0, 0, Y, errs, 1000);
    hipDeviceSynchronize();
    unsigned ab, er; hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost); hipMemcpy(&er, errs, 4, hipMemcpyDeviceToHost);
    printf("abort=%u, sum errors=%u\n", ab, er);
    return 0;
}
