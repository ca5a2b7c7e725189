// rnde_device.h -- device-side building blocks shared by the forward and reverse step kernels.
//
// Geometry ("column-owner" design):
//   one workgroup (8 waves, 512 threads) owns BT = 4*NG batch columns for a whole Tsit5 attempt.
//   The Dense layers run on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 rank-1 blocks per
//   instruction, arranged here as RG = 16/NG row-groups x NG column-groups, i.e. one instruction
//   updates a (TR = 64/NG rows) x (BT columns) tile with ONE k.  Lane l = 4*b + x, b = block:
//       A operand : W[tile*TR + (l % TR)][k]
//       B operand : X[k][4*(l / TR) + x]
//       D reg i   : out[tile*TR + 4*(b % RG) + i][4*(b / RG) + x]
//   (layout measured on gfx950 with tools/probe_mfma.hip).
//   A lane therefore owns 4 consecutive rows of one column: 16 contiguous bytes in the
//   column-major D x B state arrays, and the SAME (row, col) set for every array -- so all
//   Runge-Kutta linear combinations, the error estimate and the reverse-mode accumulators are
//   pure register arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace rnde {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

// ---- Tsit5 tableau (Tsitouras 2011), rounded to fp32 at use --------------
__host__ __device__ constexpr float tsA(int s, int j) {  // s, j zero-based: stage s+1 uses k_{j+1}
    constexpr double A[7][7] = {
        {0},
        {0.161},
        {-0.008480655492356989, 0.335480655492357},
        {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
        {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
        {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
        {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0.0}};
    return (float)A[s][j];
}
__host__ __device__ constexpr float tsC(int s) {
    constexpr double C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
    return (float)C[s];
}
__host__ __device__ constexpr float tsBt(int j) {
    constexpr double BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                              -0.1447110071732629,     0.5823571654525552,     -0.45808210592918697,
                              0.015151515151515152};
    return (float)BT[j];
}
// PI controller constants
constexpr float kBeta1 = (float)(7.0 / 50.0);
constexpr float kBeta2 = (float)(2.0 / 25.0);
constexpr float kGamma = 0.9f;
constexpr float kQmin = 0.2f;
constexpr float kQmax = 10.0f;
constexpr float kQoldInit = 1e-4f;
constexpr float kDtMin = 1.1920929e-7f;  // eps(Float32)

template <int NG>
struct Geo {
    static constexpr int BT = 4 * NG;        // batch columns per workgroup
    static constexpr int RG = 16 / NG;       // row groups per MFMA
    static constexpr int TR = 64 / NG;       // rows per MFMA tile
    static constexpr int MTS = 128 / TR;     // max tiles of the "small-M" GEMM (M <= 128)
    static constexpr int TPW = (NG == 1) ? 2 : 4;  // tiles per wave of the "big-M" GEMM (M <= 8*TPW*TR)
};

// flags in StepMeta
enum : int { F_ACCEPT = 1, F_CLAMP = 2, F_QCLAMP = 4, F_EZERO = 8, F_DTMAXCLAMP = 16, F_REJQ11 = 32 };

struct StepState {  // state BEFORE an attempt (double-buffered in HBM, one writer: workgroup 0)
    float t, dtp, qold, last_eest;
    int live;    // tape record holding the current (uprev, k1); -1 = (x, f0)
    int done;    // integration finished or aborted
    int status;  // rnde_status of the solve
    int n_att, n_acc;
    int pad[3];
};
struct StepMeta {  // one per attempted step; consumed by the reverse pass and by the host
    float t, dt, dtp_in, eest, q11, q, qold_in, rej_m;
    int flags, src, rec, pad;
};
struct InitRec {  // initial-step heuristic record
    float d0, d1, d2, dt0, dt1, dt;
    int dt0_const, dt0_clamped, sel, dt1_const, max_is_d2, pad;
};

struct StepParams {
    const float* x;  // D x B, caller layout
    float* f0; float* h0; float* u1; float* f1; float* h1;  // D x Bpad / H x Bpad
    float* arena; long long rec_stride;                      // tape records
    const f32x4* pw1; const f32x4* pw2;                      // packed weights (forward)
    StepState* ctl;     // [2]
    StepState* ctl_final;
    StepMeta* meta;
    InitRec* initrec;
    float* errpart;     // [2][nwg]
    float* initpart;    // [3][nwg]
    float* dbg_out;     // FEVAL output (caller layout) / finish copy target
    int D, H, B, Bpad, nwg;
    int K4_1, KS1, MT1, K4_2, KS2, MT2;
    float reltol, abstol, t0, t1;
    int tape, max_attempts;
    int forced; float forced_t, forced_dt;  // debug/bench: run one attempt from a given (t, dt)
    int xvec;                                // x is 16-byte aligned and D % 4 == 0
};

// record layout inside the arena (floats): k2..k7 | g2..g6 | unew | h2..h7 | z1bar2..z1bar7
// (the reverse pass overwrites k_s in place with the layer-2 pre-activation cotangent z2bar_s once
//  k_s is dead, and stores the layer-1 one in z1)
struct RecLayout {
    long long A, HB;
    __host__ __device__ long long k(int s) const { return (long long)(s - 2) * A; }       // s = 2..7
    __host__ __device__ long long g(int s) const { return (long long)(6 + s - 2) * A; }   // s = 2..6
    __host__ __device__ long long unew() const { return 11LL * A; }
    __host__ __device__ long long h(int s) const { return 12LL * A + (long long)(s - 2) * HB; }
    __host__ __device__ long long z1(int s) const { return 12LL * A + 6LL * HB + (long long)(s - 2) * HB; }
    __host__ __device__ long long total() const { return 12LL * A + 12LL * HB; }
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double wave_sum_d(double s) {
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
__device__ __forceinline__ float wave_sum_f(float s) {
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
// fixed-order sum of n fp32 partials, carried in double; identical result on every lane / workgroup
__device__ __forceinline__ double sum_partials(const float* __restrict__ part, int n, int lane) {
    double s = 0;
    for (int i = lane; i < n; i += 64) s += (double)part[i];
    return wave_sum_d(s);
}

// ---- small-M GEMM: out[M<=128][BT] = PW[M][K] * XL[K][BT]; K split over the 8 waves --------------
// pw: packed [tile][k4][TR] float4 (4 consecutive k per lane).  XL: LDS, column c at XL + c*KS.
template <int NG>
__device__ __forceinline__ void gemm_ksplit(const f32x4* __restrict__ pw, int MT, int K4, const float* XL, int KS,
                                            f32x4 (&acc)[Geo<NG>::MTS], int wave, int lane) {
    using G = Geo<NG>;
    const int chunk = (K4 + kWaves - 1) / kWaves;
    const int kb = wave * chunk;
    const int ke = min(K4, kb + chunk);
    const int ar = lane & (G::TR - 1);
    const float* xcol = XL + (4 * (lane / G::TR) + (lane & 3)) * KS;
#pragma unroll
    for (int T = 0; T < G::MTS; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 an[G::MTS];
    if (kb < ke) {
#pragma unroll
        for (int T = 0; T < G::MTS; ++T)
            if (T < MT) an[T] = pw[(size_t)(T * K4 + kb) * G::TR + ar];
    }
    for (int k4 = kb; k4 < ke; ++k4) {
        f32x4 a[G::MTS];
#pragma unroll
        for (int T = 0; T < G::MTS; ++T) a[T] = an[T];
        if (k4 + 1 < ke) {
#pragma unroll
            for (int T = 0; T < G::MTS; ++T)
                if (T < MT) an[T] = pw[(size_t)(T * K4 + k4 + 1) * G::TR + ar];
        }
        const f32x4 b = *(const f32x4*)(xcol + 4 * k4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int T = 0; T < G::MTS; ++T)
                if (T < MT) acc[T] = mfma4(a[T][kk], b[kk], acc[T]);
        }
    }
}

// ---- big-M GEMM: out[M][BT] = PW[M][K] * XL[K][BT]; tiles T = wave + 8*j owned by this wave ------
template <int NG>
__device__ __forceinline__ void gemm_rows(const f32x4* __restrict__ pw, int MT, int K4, const float* XL, int KS,
                                          f32x4 (&acc)[Geo<NG>::TPW], int wave, int lane) {
    using G = Geo<NG>;
    const int ar = lane & (G::TR - 1);
    const float* xcol = XL + (4 * (lane / G::TR) + (lane & 3)) * KS;
#pragma unroll
    for (int j = 0; j < G::TPW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 an[G::TPW];
#pragma unroll
    for (int j = 0; j < G::TPW; ++j)
        if (wave + kWaves * j < MT) an[j] = pw[(size_t)((wave + kWaves * j) * K4) * G::TR + ar];
    for (int k4 = 0; k4 < K4; ++k4) {
        f32x4 a[G::TPW];
#pragma unroll
        for (int j = 0; j < G::TPW; ++j) a[j] = an[j];
        if (k4 + 1 < K4) {
#pragma unroll
            for (int j = 0; j < G::TPW; ++j)
                if (wave + kWaves * j < MT) an[j] = pw[(size_t)((wave + kWaves * j) * K4 + k4 + 1) * G::TR + ar];
        }
        const f32x4 b = *(const f32x4*)(xcol + 4 * k4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < G::TPW; ++j)
                if (wave + kWaves * j < MT) acc[j] = mfma4(a[j][kk], b[kk], acc[j]);
        }
    }
}

// position of output (r, c) of a small-M GEMM inside the PART scratch (float index, wave 0)
template <int NG>
__device__ __forceinline__ int part_index(int r, int c) {
    using G = Geo<NG>;
    const int T = r / G::TR, rr = r % G::TR;
    const int lane = 4 * ((c >> 2) * G::RG + (rr >> 2)) + (c & 3);
    return (T * 64 + lane) * 4 + (rr & 3);
}

__device__ __forceinline__ float act_apply(int act, float v) { return act ? tanhf(v) : v; }

}  // namespace rnde
