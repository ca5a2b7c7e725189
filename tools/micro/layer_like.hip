// micro-benchmark: one "layer" = bias read + 4 interleaved chains of NK MFMAs with LDS-resident A fragments + feedback
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NK, int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float in[16];
    for (int q = 0; q < 16; ++q) in[q] = 0.5f + lane * 1e-3f + q;
    unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 acc[4];
        const float* bf = lds + 12000 + lane;
        if (MODE & 1) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc[ks >> 2][ks & 3] = bf[ks * 64];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const float* fr = lds + lane + (it & 1) * 64;
        float a[4][NK];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) a[mo][kk] = (MODE & 2) ? fr[(mo * NK + kk) * 64] : in[(kk + mo) & 15];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) acc[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mo][kk], in[kk], acc[mo], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) in[ks] = (MODE & 4) ? acc[ks >> 2][ks & 3] * 1e-3f + 0.5f : in[ks] + acc[ks >> 2][ks & 3] * 1e-9f;
    }
    unsigned long long c1 = clock64();
    float s = 0; for (int q = 0; q < 16; ++q) s += in[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = c1 - c0;
}
template <int NK, int MODE> void run(float* out, unsigned long long* st) {
    hipFuncSetAttribute((const void*)k<NK, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NK, MODE>), dim3(8), dim3(256), 65536 + 4096, 0, out, st, 1000);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("NK %2d mode %d (bias-from-LDS %d, A-from-LDS %d, full feedback %d): %.0f cycles per layer, %.1f per MFMA\n", NK, MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, h / 1000.0, h / 1000.0 / (4 * NK));
}
int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 64 * 256 * 4); hipMalloc(&st, 64 * 8);
    run<13, 0>(out, st); run<13, 1>(out, st); run<13, 2>(out, st); run<13, 3>(out, st); run<13, 7>(out, st);
    run<5, 0>(out, st); run<5, 7>(out, st);
    return 0;
}
