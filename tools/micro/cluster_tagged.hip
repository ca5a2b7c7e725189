// micro-benchmark + checker: tagged slab hand-off (validity travels inside the 16-byte entries), values verified exactly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int R = 7, C = 32, HT = 7;
__device__ __forceinline__ float val(int it, int rb, int w, int lane, int q) { return __builtin_bit_cast(float, (unsigned)((((it * 7 + rb) * 7 + w) * 64 + lane) * 4 + q) & 0x3fffffffu); }
__global__ __launch_bounds__(448) void k(float* tslab, unsigned* abort_flag, unsigned long long* st, unsigned* errs, int iters, unsigned base) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = (blockIdx.x >> 3) % R, ct = 8 * ((blockIdx.x >> 3) / R) + (blockIdx.x & 7);
    unsigned long long c0 = clock64();
    unsigned nerr = 0;
    bool dead = false;
    for (int it = 0; it < iters && !dead; ++it) {
        const int par = it & 1;
        const unsigned tag = base + it + 1;
        const float tf = __builtin_bit_cast(float, tag);
        f32x4* d = (f32x4*)tslab + ((((size_t)par * C + ct) * R + rb) * HT + w) * 128;
        d[lane] = (f32x4){val(it, rb, w, lane, 0), val(it, rb, w, lane, 1), tf, tf};
        d[64 + lane] = (f32x4){val(it, rb, w, lane, 2), val(it, rb, w, lane, 3), tf, tf};
        const float* bptr = tslab + ((((size_t)par * C + ct) * R) * HT) * 512;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)bptr, 0, 0x7fffffff, 0x00020000);
        int spins = 0;
        while (true) {
            u32x4 e0[R], e1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int off = ((r * HT + w) * 128 + lane) * 16;
                e0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                e1[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 16);
            }
            bool ok = true;
#pragma unroll
            for (int r = 0; r < R; ++r) ok = ok && e0[r][2] == tag && e0[r][3] == tag && e1[r][2] == tag && e1[r][3] == tag;
            if (__all(ok)) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (e0[r][0] != __builtin_bit_cast(unsigned, val(it, r, w, lane, 0))) ++nerr;
                    if (e0[r][1] != __builtin_bit_cast(unsigned, val(it, r, w, lane, 1))) ++nerr;
                    if (e1[r][0] != __builtin_bit_cast(unsigned, val(it, r, w, lane, 2))) ++nerr;
                    if (e1[r][1] != __builtin_bit_cast(unsigned, val(it, r, w, lane, 3))) ++nerr;
                }
                break;
            }
            if (++spins > 100000) { atomicExch(abort_flag, 1u); dead = true; break; }
        }
        __syncthreads();   // (the real kernels have workgroup barriers between a consume and the next produce)
    }
    unsigned long long c1 = clock64();
    if (nerr) atomicAdd(errs, nerr);
    if (tid == 0) st[blockIdx.x] = c1 - c0;
}
int main() {
    float* tslab; unsigned *abortf, *errs; unsigned long long* st;
    const int G = R * C;
    const size_t bytes = (size_t)2 * C * R * HT * 128 * 16;
    hipMalloc(&tslab, bytes); hipMalloc(&abortf, 4); hipMalloc(&errs, 4); hipMalloc(&st, G * 8);
    hipMemset(tslab, 0, bytes); hipMemset(abortf, 0, 4); hipMemset(errs, 0, 4);
    const int iters = 2000;
    unsigned base = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(G), dim3(448), 0, 0, tslab, abortf, st, errs, iters, base);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        base += iters;
        unsigned ab, er; hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost); hipMemcpy(&er, errs, 4, hipMemcpyDeviceToHost);
        printf("rep %d: %.3f us per hand-off (kernel %.2f ms), abort=%u, value errors=%u\n", rep, ms * 1e3 / iters, ms, ab, er);
    }
    return 0;
}
