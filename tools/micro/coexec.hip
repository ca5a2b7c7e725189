// micro-benchmark: do fp32 MFMA and fp32 VALU instructions of two waves on ONE SIMD execute together on gfx950, and is
// SQ_VALU_MFMA_COEXEC_CYCLES a live counter?  (DESIGN 5.2 read the counter's 0 as "matrix and vector issue add".)
//
// One workgroup of 512 threads per CU: waves w and w + 4 share SIMD w (MI355X_MICROARCH.md: partners are w, w + 4).  Every variant is its own
// kernel (template parameter) so that rocprofv3 --pmc reports its counters per variant:
//   MFMA_ONLY    waves 0-3: N back-to-back v_mfma_f32_16x16x4_f32 on 4 independent accumulators; waves 4-7 leave at once
//   VALU_ONLY    waves 4-7: M independent v_fma_f32 (8 chains); waves 0-3 leave at once
//   SPLIT        waves 0-3 the MFMAs, waves 4-7 the FMAs, at the same time (roles split between SIMD partners)
//   LOCKSTEP     all 8 waves run the same program: [N/K MFMAs, M/K FMAs] x K (both partners hit matrix and vector bursts together: the attempt kernels' shape)
//   STAGGER      same work per wave as LOCKSTEP, waves 4-7 start with the vector burst: [FMAs, MFMAs] x K
//   ONEWAVE_MIX  waves 0-3 alone, one MFMA followed by its share (8) of independent FMAs (does a wave's own VALU run in its MFMA's shadow?)
//   ONEWAVE_SEQ  waves 0-3 alone, the same instructions as two bursts (all MFMAs, then all FMAs): the no-overlap reference of ONEWAVE_MIX
//   TWOWAVE_MIX  all 8 waves run ONEWAVE_MIX's stream on half the work each (a SIMD carries the same totals as SPLIT)
//   SPLIT_PRIO_V / _M   SPLIT with s_setprio 3 on the vector / the matrix waves (does the arbiter let the other pipe's wave in?)
//   BF16_*       the control: MFMA_ONLY / SPLIT / LOCKSTEP with v_mfma_f32_16x16x16_bf16 (the matrix core proper) in place of the fp32 MFMA -- if THESE overlap
//                with fp32 FMAs (and the counter moves) the fp32 result above is a property of the fp32 matrix path, not of the arbiter or of the counter
//   SPLIT_NOP    SPLIT, the matrix wave issues `s_nop 7` x 3 behind every MFMA (its next MFMA does not sit at the issue stage while the pipe is busy)
// Output: shader cycles and wall time per variant; the COEXEC counter comes from the --pmc pass (tools/gpu_coexec.sh).
//   time(SPLIT) ~ max(MFMA_ONLY, VALU_ONLY): the pipes co-execute across waves;  ~ sum: they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MFMA_ONLY = 0, VALU_ONLY = 1, SPLIT = 2, LOCKSTEP = 3, STAGGER = 4, ONEWAVE_MIX = 5, SPLIT_PRIO_V = 6, SPLIT_PRIO_M = 7, SPLIT_NOP = 8, TWOWAVE_MIX = 9, ONEWAVE_SEQ = 10, BF16_MFMA_ONLY = 11, BF16_SPLIT = 12, BF16_LOCKSTEP = 13 };

// nm MFMAs on four independent accumulators (no back-to-back dependency stall: 16x16x4 f32 is 8 passes = 32 cycles, the same accumulator comes round every 4).
// Inline asm, 16 per loop trip: exactly these instructions, whatever the optimiser would make of the C form.
__device__ __forceinline__ void mfma_burst(f32x4 (&acc)[4], float a, float b, int nm) {
    for (int i = 0; i < nm; i += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
}
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_burst_bf16(f32x4 (&acc)[4], bf16x4 a, bf16x4 b, int nm) {
    for (int i = 0; i < nm; i += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
    }
}
// nv v_fma_f32 on eight independent chains, 32 per loop trip
__device__ __forceinline__ void valu_burst(float (&x)[8], float m, float c, int nv) {
    for (int i = 0; i < nv; i += 32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void coexec_kernel(float* out, unsigned long long* st, int nm, int nv, int K) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 1e-3f * (lane + j);
    const float a = 0.5f + lane * 1e-3f, b = 0.25f - lane * 1e-3f, m = 0.999f, c = 1e-4f;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    if (MODE == MFMA_ONLY) { if (w < 4) mfma_burst(acc, a, b, nm); }
    else if (MODE == VALU_ONLY) { if (w >= 4) valu_burst(x, m, c, nv); }
    else if (MODE == SPLIT) { if (w < 4) mfma_burst(acc, a, b, nm); else valu_burst(x, m, c, nv); }
    else if (MODE == LOCKSTEP) {
        for (int k = 0; k < K; ++k) { mfma_burst(acc, a, b, nm / (2 * K)); __builtin_amdgcn_sched_barrier(0); valu_burst(x, m, c, nv / (2 * K)); __builtin_amdgcn_sched_barrier(0); }
    } else if (MODE == STAGGER) {
        if (w < 4) for (int k = 0; k < K; ++k) { mfma_burst(acc, a, b, nm / (2 * K)); __builtin_amdgcn_sched_barrier(0); valu_burst(x, m, c, nv / (2 * K)); __builtin_amdgcn_sched_barrier(0); }
        else for (int k = 0; k < K; ++k) { valu_burst(x, m, c, nv / (2 * K)); __builtin_amdgcn_sched_barrier(0); mfma_burst(acc, a, b, nm / (2 * K)); __builtin_amdgcn_sched_barrier(0); }
    } else if (MODE == ONEWAVE_MIX || MODE == TWOWAVE_MIX) {      // per MFMA eight independent FMAs, same wave
        const int n = MODE == TWOWAVE_MIX ? nm / 2 : nm;
        if (MODE == TWOWAVE_MIX || w < 4) {
            for (int i = 0; i < n; i += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(m), "v"(c));
                }
            }
        }
    } else if (MODE == BF16_MFMA_ONLY || MODE == BF16_SPLIT || MODE == BF16_LOCKSTEP) {
        const bf16x4 ab = {(short)0x3f80, (short)0x3f00, (short)(0x3e80 + lane), (short)0x3f80}, bb = {(short)0x3f00, (short)0x3f80, (short)0x3e00, (short)(0x3f00 + lane)};
        if (MODE == BF16_MFMA_ONLY) { if (w < 4) mfma_burst_bf16(acc, ab, bb, nm); }
        else if (MODE == BF16_SPLIT) { if (w < 4) mfma_burst_bf16(acc, ab, bb, nm); else valu_burst(x, m, c, nv); }
        else for (int k = 0; k < K; ++k) { mfma_burst_bf16(acc, ab, bb, nm / (2 * K)); __builtin_amdgcn_sched_barrier(0); valu_burst(x, m, c, nv / (2 * K)); __builtin_amdgcn_sched_barrier(0); }
    } else if (MODE == ONEWAVE_SEQ) {
        if (w < 4) { mfma_burst(acc, a, b, nm); valu_burst(x, m, c, nv); }
    } else if (MODE == SPLIT_PRIO_V) {
        if (w < 4) mfma_burst(acc, a, b, nm); else { __builtin_amdgcn_s_setprio(3); valu_burst(x, m, c, nv); __builtin_amdgcn_s_setprio(0); }
    } else if (MODE == SPLIT_PRIO_M) {
        if (w < 4) { __builtin_amdgcn_s_setprio(3); mfma_burst(acc, a, b, nm); __builtin_amdgcn_s_setprio(0); } else valu_burst(x, m, c, nv);
    } else if (MODE == SPLIT_NOP) {
        if (w < 4) {
            for (int i = 0; i < nm; i += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
        } else valu_burst(x, m, c, nv);
    }
    __syncthreads();
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* st, int grid, int nm, int nv, int K, FILE* csv) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(coexec_kernel<MODE>, dim3(grid), dim3(512), 0, 0, out, st, nm, nv, K);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(coexec_kernel<MODE>, dim3(grid), dim3(512), 0, 0, out, st, nm, nv, K);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * grid);
    CK(hipMemcpy(h, st, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1] * 10.0; }
    cyc /= grid; wall /= grid;
    printf("%-12s grid %4d nm %6d nv %7d K %3d : %10.0f shader cycles in-kernel, %9.2f us in-kernel (100 MHz clock), %9.2f us per launch (events)\n", name, grid, nm, nv, K, cyc, wall * 1e-3,
           1e3 * ms / reps);
    if (csv) fprintf(csv, "%s,%d,%d,%d,%d,%.0f,%.3f,%.3f\n", name, grid, nm, nv, K, cyc, wall * 1e-3, 1e3 * ms / reps);
    free(h);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, (size_t)grid * 512 * 4)); CK(hipMalloc(&st, (size_t)grid * 16));
    FILE* csv = (argc > 2 && argv[2][0]) ? fopen(argv[2], "w") : nullptr;
    const int only_ratio = argc > 3 ? atoi(argv[3]) : 0, only_K = argc > 4 ? atoi(argv[4]) : 0;      // a --pmc pass wants ONE configuration per kernel name
    if (csv) fprintf(csv, "variant,grid,n_mfma_per_wave,n_fma_per_wave,K,shader_cycles_in_kernel,us_in_kernel,us_per_launch_events\n");
    // nm MFMAs x 32 cycles each; nv FMAs x 4 cycles each (one wave64 fp32 VALU instruction = 4 cycles on a 16-lane SIMD): nv = 8 nm makes the two bursts equally long
    const int nm = 4096;
    for (int ratio : {8, 4}) {
        if (only_ratio && ratio != only_ratio) continue;
        const int nv = ratio * nm;
        run<MFMA_ONLY>("MFMA_ONLY", out, st, grid, nm, nv, 1, csv);
        run<VALU_ONLY>("VALU_ONLY", out, st, grid, nm, nv, 1, csv);
        run<SPLIT>("SPLIT", out, st, grid, nm, nv, 1, csv);
        // LOCKSTEP / STAGGER: every wave does nm/2 MFMAs + nv/2 FMAs, so a SIMD (two waves) carries the same nm + nv as SPLIT
        for (int K : {64, 16, 4}) {
            if (only_K && K != only_K) continue;
            run<LOCKSTEP>("LOCKSTEP", out, st, grid, nm, nv, K, csv);
            run<STAGGER>("STAGGER", out, st, grid, nm, nv, K, csv);
        }
        if (ratio == 8) {      // (these streams are written for 8 FMAs per MFMA)
            run<ONEWAVE_SEQ>("ONEWAVE_SEQ", out, st, grid, nm, nv, 1, csv);
            run<ONEWAVE_MIX>("ONEWAVE_MIX", out, st, grid, nm, nv, 1, csv);
            run<TWOWAVE_MIX>("TWOWAVE_MIX", out, st, grid, nm, nv, 1, csv);
        }
        run<BF16_MFMA_ONLY>("BF16_MFMA_ONLY", out, st, grid, nm, nv, 1, csv);
        run<BF16_SPLIT>("BF16_SPLIT", out, st, grid, nm, nv, 1, csv);
        run<BF16_LOCKSTEP>("BF16_LOCKSTEP", out, st, grid, nm, nv, only_K ? only_K : 64, csv);
        run<SPLIT_PRIO_V>("SPLIT_PRIO_V", out, st, grid, nm, nv, 1, csv);
        run<SPLIT_PRIO_M>("SPLIT_PRIO_M", out, st, grid, nm, nv, 1, csv);
        run<SPLIT_NOP>("SPLIT_NOP", out, st, grid, nm, nv, 1, csv);
    }
    if (csv) fclose(csv);
    return 0;
}
