// micro-benchmark: dependent v_mfma_f32_16x16x4_f32 chains, LDS-fed, with 1..N workgroups -- cycles per MFMA and the
// effective shader clock (s_memtime vs the 100 MHz wall clock) when only a few CUs are busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters, int mode) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float b = 0.5f + lane * 1e-3f;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {          // pure dependent chain, A from a register
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc, 0, 0, 0);
        } else if (mode == 1) {   // A from LDS, loads batched ahead
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = lds[((it * 16 + j) & 255) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                  // result fed back as the B operand (layer-to-layer dependency) every 4 MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) t = __builtin_amdgcn_mfma_f32_16x16x4f32(b, acc[q], t, 0, 0, 0);
                acc = t * 1e-3f;
            }
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    float* out; unsigned long long* st; unsigned long long h[2];
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 16);
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {1, 8, 64, 256, 1024}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, st, 2000, mode);
            hipDeviceSynchronize();
            hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
            const double mf = 2000.0 * 16;
            printf("mode %d grid %4d: %.1f shader cycles / MFMA, %.2f ns / MFMA, shader clock %.0f MHz\n", mode, grid, h[0] / mf, h[1] * 10.0 / mf, h[0] / (h[1] * 10.0) * 1e3);
        }
    return 0;
}
