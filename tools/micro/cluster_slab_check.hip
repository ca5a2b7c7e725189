// Checker for slab_put / slab_clear / slab_poll_sum exactly as the persistent kernels use them (rnde_stage_persist.h): many launches
// of 6 exchanges each, exchange ex in buffer ex % 3, a workgroup empties its entries of buffer (ex - 1) % 3 once its poll of ex has
// succeeded and waits for those stores before its next put.  Every sum is compared with the value the producers must have sent;
// launches that exit at once (as a finished solve does) are mixed in.  Expected output: abort=0, sum errors=0.
#include "../../regneuralde.jl_amd/csrc/rnde_stage_persist.h"
#include <cstdio>
using namespace rnde;
constexpr int R = 7, C = 32, HT = 7;
__device__ __forceinline__ float val(int launch, int ex, int rb, int w, int lane, int q) {
    return (float)(((launch * 6 + ex) % 50) * 4 + rb) + 0.5f * w + 16.f * q + 64.f * (lane & 7);
}
__global__ __launch_bounds__(448) void k(PersistSync Y, unsigned* errs, int launch, int skip) {
    if (skip) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = (blockIdx.x >> 3) % R, ct = 8 * ((blockIdx.x >> 3) / R) + (blockIdx.x & 7);
    unsigned nerr = 0;
    auto put = [&](unsigned ex) {
        const size_t tile0 = (((size_t)slab_buf(ex) * C + ct) * R + rb) * HT;
        slab_put(Y.tslab, tile0 + w, lane, (f32x4){val(launch, ex, rb, w, lane, 0), val(launch, ex, rb, w, lane, 1), val(launch, ex, rb, w, lane, 2), val(launch, ex, rb, w, lane, 3)});
    };
    put(1u);
    for (unsigned ex = 1; ex <= 6; ++ex) {
        f32x4 zs;
        if (!slab_poll_sum(Y, slab_buf(ex), C, R, HT, ct, w, lane, zs)) break;
        slab_clear(Y.tslab, (((size_t)slab_buf(ex + 2u) * C + ct) * R + rb) * HT + w, lane);
        f32x4 ref = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) ref += (f32x4){val(launch, ex, r, w, lane, 0), val(launch, ex, r, w, lane, 1), val(launch, ex, r, w, lane, 2), val(launch, ex, r, w, lane, 3)};
        for (int q = 0; q < 4; ++q) if (zs[q] != ref[q]) ++nerr;
        __syncthreads();                          // the kernels' barrier between consuming an exchange and producing the next
        if (ex < 6) { slab_clears_done(); put(ex + 1u); }
    }
    if (nerr) atomicAdd(errs, nerr);
}
int main() {
    float* tslab; unsigned *abortf, *errs, *xcc;
    const int G = R * C;
    const size_t bytes = (size_t)3 * C * R * HT * 64 * 16;
    hipMalloc(&tslab, bytes); hipMalloc(&abortf, 8); hipMalloc(&errs, 4); hipMalloc(&xcc, G * 4);
    hipMemset(tslab, 0xFF, bytes); hipMemset(abortf, 0, 8); hipMemset(errs, 0, 4);
    PersistSync Y{tslab, abortf, xcc, 100000};
    for (int l = 0; l < 600; ++l) hipLaunchKernelGGL(k, dim3(G), dim3(448), 0, 0, Y, errs, l, (l % 7) == 3 ? 1 : 0);
    hipDeviceSynchronize();
    unsigned ab, er; hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost); hipMemcpy(&er, errs, 4, hipMemcpyDeviceToHost);
    printf("abort=%u, sum errors=%u\n", ab, er);
    return (ab || er) ? 1 : 0;
}
