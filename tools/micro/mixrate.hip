// mixrate.hip -- which VECTOR instructions run in the shadow of a bf16 matrix instruction of the SAME wave on gfx950?  (Round 6: the quarter-form weight-gradient
// kernel, csrc/rnde_wgradx.h, interleaves the splitting of the next step -- v_cvt_pk_bf16_f32, shifts, masks, subtractions, selects -- with the matrix instructions of
// this one and gains far less than tools/micro/coexec.hip's fp32-FMA result promises.)
// One workgroup of 8 waves per CU (two per SIMD), every wave runs  [ 1 x v_mfma_f32_16x16x32_bf16 + NV x <op> ] x N  with independent operands;
// reported: shader cycles per group and wave, for NV = 0, 2, 4, 6, 8 and <op> in { v_fma_f32, v_sub_f32, v_and_b32, v_lshlrev_b32, v_cndmask_b32, v_cvt_pk_bf16_f32,
// v_mov_b32, v_perm_b32 }.  A group costs 2 x 16 cycles of matrix pipe per SIMD (two waves): an <op> that co-executes leaves the group at ~32 cycles until its own issue
// rate binds; one that does not adds its cycles from NV = 1 on.     hipcc --offload-arch=gfx950 -O2 -o mixrate mixrate.hip && ./mixrate [csv]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define OP_FMA(x)  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define OP_SUB(x)  asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_AND(x)  asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(mi));
#define OP_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define OP_CND(x)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : );
#define OP_CVT(x)  asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MOV(x)  asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(mi));

#define KERNEL(NAME, OP, NV)                                                                                             \
    __global__ __launch_bounds__(512) void k_##NAME##_##NV(unsigned long long* out, int n) {                             \
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};                                         \
        u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {4u, 5u, 6u, threadIdx.x};                                              \
        float x[8]; for (int j = 0; j < 8; ++j) x[j] = (float)(threadIdx.x + j);                                         \
        float m = 1.0001f, c = 0.5f; unsigned mi = 0x0FFFFFFFu; (void)m; (void)mi;                                       \
        asm volatile("v_cmp_gt_u32 vcc, 32, %0" : : "v"(threadIdx.x & 63) : "vcc");                                      \
        __syncthreads();                                                                                                  \
        const unsigned long long t0 = clock64();                                                                         \
        for (int i = 0; i < n; i += 4) {                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                               \
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[r]) : "v"(a), "v"(b));                 \
                _Pragma("unroll") for (int j = 0; j < NV; ++j) { OP(x[j]) }                                               \
            }                                                                                                             \
        }                                                                                                                 \
        const unsigned long long t1 = clock64();                                                                         \
        float s = 0; for (int j = 0; j < 8; ++j) s += x[j];                                                               \
        for (int r = 0; r < 4; ++r) s += acc[r][0];                                                                       \
        if (s == 1234.5f) out[1] = 1;                                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;                                                        \
    }
#define KERNELS(NAME, OP) KERNEL(NAME, OP, 0) KERNEL(NAME, OP, 2) KERNEL(NAME, OP, 4) KERNEL(NAME, OP, 6) KERNEL(NAME, OP, 8)
KERNELS(fma, OP_FMA) KERNELS(sub, OP_SUB) KERNELS(and, OP_AND) KERNELS(lshl, OP_LSHL) KERNELS(cnd, OP_CND) KERNELS(cvt, OP_CVT) KERNELS(mov, OP_MOV) KERNELS(perm, OP_PERM)

typedef void (*kfn)(unsigned long long*, int);
#define ROW(NAME) {#NAME, {k_##NAME##_0, k_##NAME##_2, k_##NAME##_4, k_##NAME##_6, k_##NAME##_8}}
struct Row { const char* name; kfn k[5]; };
int main(int argc, char** argv) {
    const Row rows[] = {ROW(fma), ROW(sub), ROW(and), ROW(lshl), ROW(cnd), ROW(cvt), ROW(mov), ROW(perm)};
    const int nvs[5] = {0, 2, 4, 6, 8};
    const int n = 8192;
    unsigned long long* d; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "# tools/micro/mixrate.hip: 256 workgroups of 8 waves (two per SIMD), every wave [1 v_mfma_f32_16x16x32_bf16 + NV x op] x %d; shader cycles per group and wave\nop,NV,cycles_per_group\n", n);
    printf("%-8s", "op");
    for (int j = 0; j < 5; ++j) printf("  NV=%d   ", nvs[j]);
    printf("  (cycles per [MFMA + NV ops] group per wave; two waves per SIMD)\n");
    for (const Row& r : rows) {
        printf("%-8s", r.name);
        for (int j = 0; j < 5; ++j) {
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(r.k[j], dim3(256), dim3(512), 0, 0, d, n);
                CK(hipDeviceSynchronize());
                unsigned long long c; CK(hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost));
                if (c < best) best = c;
            }
            printf("  %7.2f", (double)best / n);
            if (csv) fprintf(csv, "%s,%d,%.3f\n", r.name, nvs[j], (double)best / n);
        }
        printf("\n");
    }
    if (csv) fclose(csv);
    return 0;
}
