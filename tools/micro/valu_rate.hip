// micro-benchmark: how fast does ONE wave issue vector instructions on gfx950, by ENCODING, and how does that scale with the waves on a SIMD?
// (tools/micro/coexec.hip found one wave alone at 5.8 cycles per v_fma_f32 and two waves on a SIMD at 2.9 each: the stages' element-wise phases --
//  two tanh per stage, ~200 vector instructions per wave -- run at the per-wave ISSUE rate, not at the ALU's.)
// Streams of 8 independent chains, 64 instructions per loop trip, inline asm so that the encodings are what is written here:
//   FMA_VOP3   v_fma_f32        (64-bit VOP3 encoding)
//   FMAC_VOP2  v_fmac_f32_e32   (32-bit VOP2 encoding, same operation)
//   MUL_VOP2   v_mul_f32_e32
//   PK_FMA     v_pk_fma_f32     (64-bit VOP3P, two fp32 per lane)
//   PK_MUL     v_pk_mul_f32
//   EXP        v_exp_f32_e32    (transcendental, quarter rate)
//   MOV        v_mov_b32_e32
// with 1, 2, 3, 4 waves per SIMD (workgroups of 256 x k threads, one per CU).  Output: shader cycles per instruction PER WAVE and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { FMA_VOP3 = 0, FMAC_VOP2 = 1, MUL_VOP2 = 2, PK_FMA = 3, PK_MUL = 4, EXP = 5, MOV = 6 };

template <int KIND>
__global__ __launch_bounds__(1024) void valu_kernel(float* out, unsigned long long* st, int n) {
    const int lane = threadIdx.x & 63;
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = 1.0f + 1e-3f * (lane + j); y[j] = (f32x2){x[j], x[j] + 0.5f}; }
    float m = 0.9999f, c = 1e-5f;
    f32x2 m2 = {m, m}, c2 = {c, c};
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int i = 0; i < n; i += 64) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (KIND == FMA_VOP3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
                else if (KIND == FMAC_VOP2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
                else if (KIND == MUL_VOP2) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[j]) : "v"(m));
                else if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[j]) : "v"(m2), "v"(c2));
                else if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[j]) : "v"(m2));
                else if (KIND == EXP) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[j]));
                else asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x[j]) : "v"(m));
            }
        }
    }
    __syncthreads();
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + y[j].x + y[j].y;
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = c1 - c0;
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* st, FILE* csv) {
    const int grid = 256, n = 16384;
    for (int wps = 1; wps <= 4; ++wps) {
        hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256 * wps), 0, 0, out, st, n);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256 * wps), 0, 0, out, st, n);
        CK(hipDeviceSynchronize());
        unsigned long long h[256];
        CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (int i = 0; i < grid; ++i) cyc += (double)h[i];
        cyc /= grid;
        printf("%-10s %d wave(s) per SIMD: %6.2f cycles per instruction per wave, %5.2f per SIMD\n", name, wps, cyc / n, cyc / n / wps);
        if (csv) fprintf(csv, "%s,%d,%.3f,%.3f\n", name, wps, cyc / n, cyc / n / wps);
    }
}

int main(int argc, char** argv) {
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, (size_t)256 * 1024 * 4)); CK(hipMalloc(&st, 256 * 8));
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "instruction,waves_per_simd,cycles_per_instruction_per_wave,cycles_per_instruction_per_simd\n");
    run<FMA_VOP3>("FMA_VOP3", out, st, csv);
    run<FMAC_VOP2>("FMAC_VOP2", out, st, csv);
    run<MUL_VOP2>("MUL_VOP2", out, st, csv);
    run<PK_FMA>("PK_FMA", out, st, csv);
    run<PK_MUL>("PK_MUL", out, st, csv);
    run<EXP>("EXP", out, st, csv);
    run<MOV>("MOV", out, st, csv);
    if (csv) fclose(csv);
    return 0;
}
