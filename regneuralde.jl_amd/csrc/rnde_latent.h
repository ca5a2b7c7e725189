// rnde_latent.h -- the latent-ODE caller of the hot path on the device: recognition GRU, rec_to_gen + sampling,
// gen_to_data + masked likelihood + KL, and their reverse passes.
//
// Reference: experiments/latent_ode.jl:39-106 (LatentGRU, 49 steps backwards over the time axis), :112 (rec_to_gen), :148 (gen_to_data),
// :192-204 (log_likelihood, kl_divergence), :226-236 (loss), src/models/time_series.jl:40-70 (LatentTimeSeriesModel call) and what
// Tracker.gradient computes over them (latent_ode.jl:339-347).  Shapes are the reference's (LatentGRU(37, 40, 50), Dense(100, 50, tanh) ->
// Dense(50, 40), Dense(20, 37)); they are compile-time constants of these kernels.
//
// Decomposition (one launch each):
//   rnde_latent_gru_fwd_kernel   all T recurrent steps; one workgroup (8 waves) per 16 batch columns, the six weight matrices register
//                                stationary as MFMA A fragments, activations in LDS as packed B operands, four barriers per step; every
//                                layer input / output of every step goes to the ACT tape ([sample][feature], sample = t * B + b)
//   rnde_latent_gru_bwd_kernel   the reverse recurrence: only the cotangent PROPAGATION (four products with transposed weights per step);
//                                the pre-activation cotangents go to the DEL tape
//   rnde_latent_gru_wgrad_kernel every weight gradient of the model is a GEMM  sum_samples delta^T act  over a tape pair: all jobs of a pair in one
//                                pass over whole records; per-workgroup partials summed in a fixed order (deterministic, no float atomics)
//   rnde_latent_enc_*            Dense(100, 50, tanh) -> Dense(50, 40), z0 = eps * exp(logvar / 2) + mu0, KL per sample (+ reverse)
//   rnde_latent_dec_loss_kernel  Dense(20, 37) on every saved state, masked Gaussian log likelihood / observed count, and the reverse of both
// Layouts: Julia's F x T x B arrays as they are (feature fastest): element (f, t, b) at (b * T + t) * F + f.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnde_lat {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a quad at a 4-byte aligned address (the std halves start at row 50)

constexpr int kIn = 37, kNX = 2 * kIn + 1, kH = 40, kL = 50, kNIn = 2 * kL + kNX;      // 75 input rows, 175 rows into every gate stack
constexpr int kRec = 50, kLat = 20;
// parameter offsets inside p1 (Flux.destructure of LatentGRU: update_gate, reset_gate, new_state; each Dense as [vec(W) (out x in, column-major); b])
constexpr int kGate = kNIn * kH + kH + kH * kL + kL;                 // 9090
constexpr int oWu1 = 0, obu1 = kNIn * kH, oWu2 = obu1 + kH, obu2 = oWu2 + kH * kL;
constexpr int oWr1 = kGate, obr1 = oWr1 + kNIn * kH, oWr2 = obr1 + kH, obr2 = oWr2 + kH * kL;
constexpr int oWn1 = 2 * kGate, obn1 = oWn1 + kNIn * kH, oWn2 = obn1 + kH, obn2 = oWn2 + kH * 2 * kL;
constexpr int kP1 = obn2 + 2 * kL;                                   // 29,320
constexpr int kP2 = 2 * kL * kRec + kRec + kRec * 2 * kLat + 2 * kLat;   // 7,090
constexpr int kP4 = kLat * kIn + kIn;                                // 777

// ACT tape record (floats per sample; every offset a multiple of 4: 16-byte stores from the MFMA result layout)
constexpr int aYC = 0, aCC = 176, aU1 = 352, aR1 = 392, aN1 = 432, aU = 472, aR = 524, aNSM = 576, aNSS = 628, kActLd = 680;
// DEL tape record: pre-activation cotangents (zu, zr, zn: first layers; au, ar: gate outputs; nsb: new-state output, mean at 0, std at 52)
constexpr int dZU = 0, dZR = 40, dZN = 80, dAU = 120, dAR = 172, dNS = 224, kDelLd = 328;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// packed B-operand image in LDS: element (k, col) of a [K][16] activation block; a lane (kk = lane >> 4, col = lane & 15) reads the float4
// at (g * 64 + lane) * 4 and gets the operands of the k-steps 4 g .. 4 g + 3 (k = 16 g + 4 q + kk)
__device__ __forceinline__ int pb_off(int k, int col) { return (((k >> 4) * 64 + (k & 3) * 16 + col) << 2) + ((k >> 2) & 3); }

// tanh as the stage engine evaluates it (rnde_device.h tanh_fast: odd polynomial below 0.55, 1 - 2 / (exp(2|x|) + 1) above; 1.65 ulp max)
__device__ __forceinline__ float tanh_f(float x) {
    const float ax = fabsf(x), x2 = x * x;
    float p = -0.00671552f;
    p = fmaf(p, x2, 0.02136713f); p = fmaf(p, x2, -0.05391917f); p = fmaf(p, x2, 0.13333165f); p = fmaf(p, x2, -0.33333332f);
    const float small = fmaf(x, x2 * p, x);
    constexpr float Lh = 2.8853900817779268f, Llo = (float)(2.8853900817779268 - (double)Lh);
    const float yh = ax * Lh, yl = fmaf(ax, Lh, -yh) + ax * Llo;
    float e = __builtin_amdgcn_exp2f(yh);
    e = fmaf(e, yl * 0.6931471805599453f, e);
    const float dd = e + 1.0f;
    float r = __builtin_amdgcn_rcpf(dd);
    r = fmaf(fmaf(-dd, r, 1.0f), r, r);
    float big = fmaf(-2.0f, r, 1.0f);
    big = ax > 9.1f ? 1.0f : big;
    return ax < 0.55f ? small : copysignf(big, x);
}
// 1 / (1 + exp(-x)): exp2 with the argument split as above, one Newton step on the reciprocal
__device__ __forceinline__ float sigmoid_f(float x) {
    constexpr float Lh = 1.4426950408889634f, Llo = (float)(1.4426950408889634 - (double)Lh);
    const float a = fminf(fmaxf(-x, -87.f), 87.f);
    const float yh = a * Lh, yl = fmaf(a, Lh, -yh) + a * Llo;
    float e = __builtin_amdgcn_exp2f(yh);
    e = fmaf(e, yl * 0.6931471805599453f, e);
    const float dd = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(dd);
    return fmaf(fmaf(-dd, r, 1.0f), r, r);
}

// acc = sum over k-steps [0, NKS) of A-fragment a[ks] * B image (packed, G groups): NKS <= 4 * groups
template <int NKS>
__device__ __forceinline__ f32x4 dense_tile(const float (&a)[NKS], const float* __restrict__ bimg, int lane) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < (NKS + 3) / 4; ++g) {
        const f32x4 b = *(const f32x4*)(bimg + (g * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * g + q < NKS) {
                if (q & 1) acc1 = mfma16(a[4 * g + q], b[q], acc1);
                else acc0 = mfma16(a[4 * g + q], b[q], acc0);
            }
        }
    }
    return acc0 + acc1;
}

struct GruParams {
    const float* x;        // kNX x T x B
    const float* p1;
    float* act;            // [T * B][kActLd]
    float* del;            // [T * B][kDelLd]   (backward)
    float* y;              // 2 kL x B: forward: output; backward: its cotangent (input)
    int B, T;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: T steps, t = T - 1 .. 0 (latent_ode.jl:99-106)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void rnde_latent_gru_fwd_kernel(const GruParams Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* YC = smem;                 // 11 groups
    float* CC = YC + 11 * 256;        // 11 groups
    float* UB = CC + 11 * 256;        // 3 groups (U1, 40 rows)
    float* RB = UB + 3 * 256;
    float* NB = RB + 3 * 256;
    float* MK = NB + 3 * 256;         // [T][16] step masks (latent_ode.jl:91)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q4 = lane >> 4;
    const int b0 = blockIdx.x * 16, b = b0 + col;
    const bool bok = b < Q.B;
    const float* __restrict__ p = Q.p1;

    // ---- weights: register-stationary A fragments (lane: row = tile * 16 + (lane & 15), k = 4 ks + (lane >> 4)) ----
    float wP1[44], wP2u[10], wP2r[10], wP4m[10], wP4s[10];        // P1: [U1; R1] tile w (waves 0..4), N1 tile w - 5 (waves 5..7) share wP1
    float bias1[4], bias2u[4], bias2r[4], bias4m[4], bias4s[4];
    {
        const int arow = lane & 15;
#pragma unroll
        for (int ks = 0; ks < 44; ++ks) {
            const int k = 4 * ks + q4;
            float v = 0.f;
            if (k < kNIn) {
                if (w < 5) { const int R = 16 * w + arow; v = R < kH ? p[oWu1 + k * kH + R] : p[oWr1 + k * kH + (R - kH)]; }
                else { const int R = 16 * (w - 5) + arow; if (R < kH) v = p[oWn1 + k * kH + R]; }
            }
            wP1[ks] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = 16 * (w < 5 ? w : w - 5) + 4 * q4 + i;
            bias1[i] = w < 5 ? (R < kH ? p[obu1 + R] : p[obr1 + (R - kH)]) : (R < kH ? p[obn1 + R] : 0.f);
        }
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            const int k = 4 * ks + q4, R = 16 * (w & 3) + arow;      // (waves 0..3 use them)
            const bool ok = R < kL;
            wP2u[ks] = ok ? p[oWu2 + k * kL + R] : 0.f; wP2r[ks] = ok ? p[oWr2 + k * kL + R] : 0.f;
            wP4m[ks] = ok ? p[oWn2 + k * 2 * kL + R] : 0.f; wP4s[ks] = ok ? p[oWn2 + k * 2 * kL + kL + R] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = 16 * (w & 3) + 4 * q4 + i;
            const bool ok = R < kL;
            bias2u[i] = ok ? p[obu2 + R] : 0.f; bias2r[i] = ok ? p[obr2 + R] : 0.f;
            bias4m[i] = ok ? p[obn2 + R] : 0.f; bias4s[i] = ok ? p[obn2 + kL + R] : 0.f;
        }
    }
    // ---- step masks of this tile: sum over rows kNX / 2 .. kNX - 1 of x[:, t, b] > 0 ----
    for (int i = tid; i < Q.T * 16; i += 512) {
        const int t = i >> 4, c = i & 15;
        float s = 0.f;
        if (b0 + c < Q.B) { const float* xp = Q.x + ((size_t)(b0 + c) * Q.T + t) * kNX; for (int f = kNX / 2; f < kNX; ++f) s += xp[f]; }
        MK[i] = s > 0.f ? 1.f : 0.f;
    }
    for (int i = tid; i < 22 * 256 + 9 * 256; i += 512) smem[i] = 0.f;      // padding rows of the operand images stay zero
    f32x4 ym = {0.f, 0.f, 0.f, 0.f}, ys = {0.f, 0.f, 0.f, 0.f};            // waves 0..3: rows 16 w + 4 q4 + i of y_mean / y_std
    const int yrow = 16 * (w & 3) + 4 * q4;
    const bool full = yrow + 3 < kL, part = yrow < kL;                     // (50 rows: the tile of wave 3 ends with rows 48, 49)
    // x_t: element i = tid + 512 j of the 16 x 75 block (column c, row f): loop invariants of the time loop, and the NEXT step's values are
    // requested one step ahead (the load is a cold global read: on the step's critical path otherwise)
    int xc[3], xf[3]; bool xok[3]; float xv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = tid + 512 * j;
        xc[j] = i / kNX; xf[j] = i - xc[j] * kNX; xok[j] = i < kNX * 16 && b0 + xc[j] < Q.B;
        xv[j] = xok[j] ? Q.x[((size_t)(b0 + xc[j]) * Q.T + (Q.T - 1)) * kNX + xf[j]] : 0.f;
    }
    __syncthreads();

    for (int t = Q.T - 1; t >= 0; --t) {
        const size_t smp = (size_t)t * Q.B + b;
        float* rec = Q.act + smp * kActLd;
        // ---- a. operand images: [y_mean; y_std; x_t] -> YC, x_t -> CC; tape of y ----
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (tid + 512 * j < kNX * 16) {
                const float v = xv[j];
                YC[pb_off(2 * kL + xf[j], xc[j])] = v; CC[pb_off(2 * kL + xf[j], xc[j])] = v;
                if (xok[j]) { float* rc = Q.act + ((size_t)t * Q.B + b0 + xc[j]) * kActLd; rc[aYC + 2 * kL + xf[j]] = v; rc[aCC + 2 * kL + xf[j]] = v; }
                if (t > 0) xv[j] = xok[j] ? Q.x[((size_t)(b0 + xc[j]) * Q.T + (t - 1)) * kNX + xf[j]] : 0.f;
            }
        }
        if (w < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (yrow + i < kL) { YC[pb_off(yrow + i, col)] = ym[i]; YC[pb_off(kL + yrow + i, col)] = ys[i]; }
            if (bok && part) {      // (16-byte tape stores; rows 50, 51 of the last tile land in the padding in front of the next array's first rows --
                if (full) { *(f32x4*)(rec + aYC + yrow) = ym; *(f32x4u*)(rec + aYC + kL + yrow) = ys; }      //  which are written later in the step, so only full quads go out as vectors)
                else { rec[aYC + yrow] = ym[0]; rec[aYC + yrow + 1] = ym[1]; rec[aYC + kL + yrow] = ys[0]; rec[aYC + kL + yrow + 1] = ys[1]; }
            }
        }
        __syncthreads();
        // ---- b. P1: [U1; R1] = tanh(W1 yc + b1), 80 rows on waves 0..4 ----
        if (w < 5) {
            f32x4 z = dense_tile<44>(wP1, YC, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = 16 * w + 4 * q4 + i;
                const float v = tanh_f(z[i] + bias1[i]);
                if (R < kH) { UB[pb_off(R, col)] = v; if (bok) rec[aU1 + R] = v; }
                else if (R < 2 * kH) { RB[pb_off(R - kH, col)] = v; if (bok) rec[aR1 + R - kH] = v; }
            }
        }
        __syncthreads();
        // ---- c. P2 on waves 0..3: u = sigmoid(Wu2 U1 + b), r = sigmoid(Wr2 R1 + b); concat = [y_mean r; y_std r; x] ----
        f32x4 u = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
        if (w < 4) {
            const f32x4 zu = dense_tile<10>(wP2u, UB, lane), zr = dense_tile<10>(wP2r, RB, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = sigmoid_f(zu[i] + bias2u[i]); r[i] = sigmoid_f(zr[i] + bias2r[i]);
                if (yrow + i < kL) {
                    const float cm = ym[i] * r[i], cs = ys[i] * r[i];
                    CC[pb_off(yrow + i, col)] = cm; CC[pb_off(kL + yrow + i, col)] = cs;
                    if (bok && !full) { rec[aCC + yrow + i] = cm; rec[aCC + kL + yrow + i] = cs; }
                }
            }
            if (bok && part) {      // (u, r have 52-float slots: whole quads always)
                *(f32x4*)(rec + aU + yrow) = u; *(f32x4*)(rec + aR + yrow) = r;
                if (full) { *(f32x4*)(rec + aCC + yrow) = ym * r; *(f32x4u*)(rec + aCC + kL + yrow) = ys * r; }
            }
        }
        __syncthreads();
        // ---- d. P3 on waves 5..7: N1 = tanh(Wn1 concat + b) ----
        if (w >= 5) {
            f32x4 z = dense_tile<44>(wP1, CC, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = 16 * (w - 5) + 4 * q4 + i;
                const float v = tanh_f(z[i] + bias1[i]);
                if (R < kH) { NB[pb_off(R, col)] = v; if (bok) rec[aN1 + R] = v; }
            }
        }
        __syncthreads();
        // ---- e. P4 on waves 0..3: new state (mean, std rows of this tile) and the gated, masked update ----
        if (w < 4) {
            const f32x4 nm = dense_tile<10>(wP4m, NB, lane), ns = dense_tile<10>(wP4s, NB, lane);
            const float m = MK[t * 16 + col];
            f32x4 vm4, vs4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float vm = nm[i] + bias4m[i], vs = ns[i] + bias4s[i];
                vm4[i] = vm; vs4[i] = vs;
                const float nym = (1.f - u[i]) * vm + u[i] * ym[i], nys = (1.f - u[i]) * vs + u[i] * ys[i];
                ym[i] = m * nym + (1.f - m) * ym[i]; ys[i] = m * nys + (1.f - m) * ys[i];
            }
            if (bok && part) { *(f32x4*)(rec + aNSM + yrow) = vm4; *(f32x4*)(rec + aNSS + yrow) = vs4; }      // (52-float slots)
        }
        // (the next step's image writes touch YC / CC rows that nobody reads any more: phase d was behind the last barrier)
    }
    if (w < 4 && bok) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (yrow + i < kL) { Q.y[(size_t)b * 2 * kL + yrow + i] = ym[i]; Q.y[(size_t)b * 2 * kL + kL + yrow + i] = ys[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward: the steps in the reverse of their forward order (t = 0 .. T - 1); cotangent propagation only
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void rnde_latent_gru_bwd_kernel(const GruParams Q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* NSB = smem;                // 7 groups: new-state cotangent, mean rows 0..49, std rows 52..101
    float* ZNB = NSB + 7 * 256;       // 3 groups
    float* AUB = ZNB + 3 * 256;       // 4 groups
    float* ARB = AUB + 4 * 256;       // 4 groups
    float* ZUR = ARB + 4 * 256;       // 5 groups: zu rows 0..39, zr rows 40..79
    float* MK = ZUR + 5 * 256;        // [T][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q4 = lane >> 4, arow = lane & 15;
    const int b0 = blockIdx.x * 16, b = b0 + col;
    const bool bok = b < Q.B;
    const float* __restrict__ p = Q.p1;

    // transposed weights as A fragments:
    //   tA (waves 5..7): Wn2^T rows (hidden j = 16 (w - 5) + arow), k over the new-state rows in the NSB numbering (26 k-steps)
    //   tB (waves 0..3): Wn1^T rows (mean row, std row of tile w), k over the 40 hidden units (10 k-steps each)
    //   tC (waves 0..5): Wu2^T / Wr2^T rows (hidden j), k over the 50 gate rows in the AUB numbering (13 k-steps)
    //   tD (waves 0..3): [Wu1^T Wr1^T] rows (mean row, std row of tile w), k over [zu; zr] (20 k-steps each)
    float tA[26], tBm[10], tBs[10], tC[13], tDm[20], tDs[20];
#pragma unroll
    for (int ks = 0; ks < 26; ++ks) {
        const int k = 4 * ks + q4, j = 16 * (w >= 5 ? w - 5 : 0) + arow;
        float v = 0.f;
        if (w >= 5 && j < kH) { if (k < kL) v = p[oWn2 + j * 2 * kL + k]; else if (k >= 52 && k < 52 + kL) v = p[oWn2 + j * 2 * kL + kL + (k - 52)]; }
        tA[ks] = v;
    }
    const int mrow = 16 * (w & 3) + arow;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        const int k = 4 * ks + q4;      // hidden unit
        const bool ok = mrow < kL;
        tBm[ks] = ok ? p[oWn1 + mrow * kH + k] : 0.f; tBs[ks] = ok ? p[oWn1 + (kL + mrow) * kH + k] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 13; ++ks) {
        const int k = 4 * ks + q4, j = 16 * (w % 3) + arow;      // waves 0..2: Wu2^T, 3..5: Wr2^T
        tC[ks] = (w < 6 && j < kH && k < kL) ? p[(w < 3 ? oWu2 : oWr2) + j * kL + k] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) {
        const int k = 4 * ks + q4;      // 0..39: zu, 40..79: zr
        const bool ok = mrow < kL;
        const int o = k < kH ? oWu1 : oWr1, kk = k < kH ? k : k - kH;
        tDm[ks] = ok ? p[o + mrow * kH + kk] : 0.f; tDs[ks] = ok ? p[o + (kL + mrow) * kH + kk] : 0.f;
    }
    for (int i = tid; i < Q.T * 16; i += 512) {
        const int t = i >> 4, c = i & 15;
        float s = 0.f;
        if (b0 + c < Q.B) { const float* xp = Q.x + ((size_t)(b0 + c) * Q.T + t) * kNX; for (int f = kNX / 2; f < kNX; ++f) s += xp[f]; }
        MK[i] = s > 0.f ? 1.f : 0.f;
    }
    for (int i = tid; i < 23 * 256; i += 512) smem[i] = 0.f;
    const int yrow = 16 * (w & 3) + 4 * q4;
    f32x4 ymb = {0.f, 0.f, 0.f, 0.f}, ysb = {0.f, 0.f, 0.f, 0.f};
    if (w < 4 && bok) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (yrow + i < kL) { ymb[i] = Q.y[(size_t)b * 2 * kL + yrow + i]; ysb[i] = Q.y[(size_t)b * 2 * kL + kL + yrow + i]; }
    }
    __syncthreads();

    // tape operands of a step, as this lane needs them (quads of its own rows): requested ONE STEP AHEAD -- they are cold global reads
    const bool part = yrow < kL;
    const int nrow = 16 * (w >= 5 ? w - 5 : 0) + 4 * q4;      // waves 5..7: their N1 rows (phase 2)
    const int hrow = 16 * (w % 3) + 4 * q4;                  // waves 0..2 / 3..5: their U1 / R1 rows (phase 4; wave 5 has both roles)
    struct StepOps { f32x4 ym, ys, u, r, nsm, nss, h1, n1; };
    auto fetch = [&](int t) {
        StepOps o;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        o.ym = o.ys = o.u = o.r = o.nsm = o.nss = o.h1 = o.n1 = z;
        if (bok && t < Q.T) {
            const float* rec = Q.act + ((size_t)t * Q.B + b) * kActLd;
            if (w < 4 && part) {
                o.u = *(const f32x4*)(rec + aU + yrow); o.r = *(const f32x4*)(rec + aR + yrow);
                o.nsm = *(const f32x4*)(rec + aNSM + yrow); o.nss = *(const f32x4*)(rec + aNSS + yrow);
                if (yrow + 3 < kL) { o.ym = *(const f32x4*)(rec + aYC + yrow); o.ys = *(const f32x4u*)(rec + aYC + kL + yrow); }
                else { o.ym[0] = rec[aYC + yrow]; o.ym[1] = rec[aYC + yrow + 1]; o.ys[0] = rec[aYC + kL + yrow]; o.ys[1] = rec[aYC + kL + yrow + 1]; }
            }
            if (w < 6 && hrow < kH) o.h1 = *(const f32x4*)(rec + (w < 3 ? aU1 : aR1) + hrow);
            if (w >= 5 && nrow < kH) o.n1 = *(const f32x4*)(rec + aN1 + nrow);
        }
        return o;
    };
    StepOps nx = fetch(0);
    for (int t = 0; t < Q.T; ++t) {
        const size_t smp = (size_t)t * Q.B + b;
        float* drec = Q.del + smp * kDelLd;
        const StepOps op = nx;
        nx = fetch(t + 1);
        const f32x4 ym = op.ym, ys = op.ys, u = op.u, r = op.r;
        f32x4 ub = {0.f, 0.f, 0.f, 0.f}, ymo = ub, yso = ub;
        // ---- 1. the gated, masked update in reverse (waves 0..3) ----
        if (w < 4) {
            const float m = MK[t * 16 + col];
            f32x4 bm4, bs4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = yrow + i;
                const float gm = m * ymb[i], gs = m * ysb[i];
                ymo[i] = (1.f - m) * ymb[i] + u[i] * gm; yso[i] = (1.f - m) * ysb[i] + u[i] * gs;
                ub[i] = (ym[i] - op.nsm[i]) * gm + (ys[i] - op.nss[i]) * gs;
                const float bm = (1.f - u[i]) * gm, bs = (1.f - u[i]) * gs;
                bm4[i] = R < kL ? bm : 0.f; bs4[i] = R < kL ? bs : 0.f;
                if (R < kL) { NSB[pb_off(R, col)] = bm; NSB[pb_off(52 + R, col)] = bs; }
            }
            if (bok && part) { *(f32x4*)(drec + dNS + yrow) = bm4; *(f32x4*)(drec + dNS + 52 + yrow) = bs4; }      // (52-float halves: whole quads, zeros in the padding)
        }
        __syncthreads();
        // ---- 2. waves 5..7: N1-bar = Wn2^T ns-bar; zn = N1-bar (1 - N1^2) ----
        if (w >= 5) {
            const f32x4 nb = dense_tile<26>(tA, NSB, lane);
            f32x4 z4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float n1 = op.n1[i];
                z4[i] = nb[i] * (1.f - n1 * n1);
                if (nrow + i < kH) ZNB[pb_off(nrow + i, col)] = z4[i];
            }
            if (bok && nrow < kH) *(f32x4*)(drec + dZN + nrow) = z4;      // (40 = 10 quads: a lane's rows are all inside or all outside)
        }
        __syncthreads();
        // ---- 3. waves 0..3: concat-bar (mean / std rows) = Wn1^T zn; r-bar, gate pre-activation cotangents ----
        if (w < 4) {
            const f32x4 cm = dense_tile<10>(tBm, ZNB, lane), cs = dense_tile<10>(tBs, ZNB, lane);
            f32x4 au4, ar4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = yrow + i;
                const float rb = ym[i] * cm[i] + ys[i] * cs[i];
                ymo[i] += r[i] * cm[i]; yso[i] += r[i] * cs[i];
                const float au = ub[i] * u[i] * (1.f - u[i]), ar = rb * r[i] * (1.f - r[i]);
                au4[i] = R < kL ? au : 0.f; ar4[i] = R < kL ? ar : 0.f;
                if (R < kL) { AUB[pb_off(R, col)] = au; ARB[pb_off(R, col)] = ar; }
            }
            if (bok && part) { *(f32x4*)(drec + dAU + yrow) = au4; *(f32x4*)(drec + dAR + yrow) = ar4; }
        }
        __syncthreads();
        // ---- 4. waves 0..2: U1-bar = Wu2^T au -> zu; waves 3..5: R1-bar = Wr2^T ar -> zr ----
        if (w < 6) {
            const f32x4 hb = dense_tile<13>(tC, w < 3 ? AUB : ARB, lane);
            const int hr = 16 * (w % 3) + 4 * q4;
            f32x4 z4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h1 = op.h1[i];
                z4[i] = hb[i] * (1.f - h1 * h1);
                if (hr + i < kH) ZUR[pb_off((w < 3 ? 0 : kH) + hr + i, col)] = z4[i];
            }
            if (bok && hr < kH) *(f32x4*)(drec + (w < 3 ? dZU : dZR) + hr) = z4;
        }
        __syncthreads();
        // ---- 5. waves 0..3: y_concat-bar (mean / std rows) = Wu1^T zu + Wr1^T zr; the cotangent of the previous state ----
        if (w < 4) {
            const f32x4 dm = dense_tile<20>(tDm, ZUR, lane), ds = dense_tile<20>(tDs, ZUR, lane);
            ymb = ymo + dm; ysb = yso + ds;
        }
        // (phase 1 of the next step writes NSB, last read in phase 2: two barriers back)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradients: out[m][n] = sum over samples s of delta[s][m] * act[s][n]   (m < M, n < N; n == N: the bias, act = 1)
// delta, act: columns of two tapes of records ([sample][features]); out: element (m, n) at n * M + m, the layout of a Flux Dense [vec(W); b]
// ---------------------------------------------------------------------------------------------------------------------------------
struct WgradJob { const float* delta; const float* act; float* out; int M, N, m_split, m_gap; };
// m_split / m_gap: output rows m >= m_split read delta column m + m_gap (the new-state cotangent keeps its std half at offset 52)
struct WgradJobs { WgradJob j[8]; int n; };

// ---------------------------------------------------------------------------------------------------------------------------------
// The weight gradients of a tape pair in ONE pass (round 4; the first form, one launch per job re-reading its operands, read the GRU's activation
// record three times: 250 MB for 100 MB of tape, and took 318 us per step for the model's nine gradients against 150 now).  A workgroup
// brings 16 whole records of both tapes into LDS (contiguous in memory: 252 `global_load_lds` requests of 256 B, 64.5 KB), and every job's
// tiles are multiplied out of that image -- 144 accumulator tiles of 16 x 16, 18 per wave, register resident across the workgroup's records.
// One workgroup of 8 waves per CU with TWO images (129 KB): the requests of the next 16 records are in flight while the current ones are multiplied
// (measured on the first, single-image form with two 4-wave workgroups per CU: 20 us fixed, + 13.5 us requests, + 26 us products, not overlapping;
// its 512 accumulator images were 75 MB of partials).  A workgroup leaves its accumulators as they
// are (16-byte stores); rnde_latent_reduce_raw_kernel / _scatter_kernel sum them over the workgroups in a fixed order (deterministic) and place
// them in the jobs' outputs.  The jobs' delta / act pointers must lie inside one DEL / ACT record (offsets are taken from them).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kFwSamples = 16;                                          // records per chunk
constexpr int kFwWaves = 8, kFwTilesPerWave = 18;                       // 144 tiles at most (the GRU's six jobs)
// act / del: first record of the two tapes, LDA / LDD floats per record (multiples of 4: compile-time, the k-steps of a tile are then immediate
// offsets of its LDS reads); the jobs' pointers lie inside the first record.  The image is followed by 16 zero floats: a lane whose row / column is
// padding of its tile reads whatever lies next to its job's columns -- that pollutes accumulator rows / columns that are never stored.
struct FusedWgrad { WgradJobs J; const float* act; const float* del; int K; float* raw; };      // raw: [workgroup][tile][64 lanes][4] accumulator images
__host__ __device__ inline int fused_tile_count(const WgradJobs& J) {
    int n = 0;
    for (int j = 0; j < J.n; ++j) n += ((J.j[j].M + 15) >> 4) * ((J.j[j].N + 1 + 15) >> 4);
    return n;
}
template <int LDA, int LDD>
__global__ __launch_bounds__(64 * kFwWaves) void rnde_latent_gru_wgrad_kernel(const FusedWgrad F) {
    extern __shared__ __attribute__((aligned(16))) float S0[];
    __shared__ int TT[kFwWaves * kFwTilesPerWave];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nA = kFwSamples * LDA, nD = kFwSamples * LDD;          // floats of a chunk's two images (multiples of 64)
    static_assert(nA + nD + 16 < (1 << 15), "record sizes");
    constexpr int kImg = nA + nD + 16;                                   // one image: both tapes' records + 16 zero floats
    constexpr bool kDma = nA % 64 == 0 && nD % 64 == 0;                  // whole 256-byte requests (records of a multiple of 4 floats); else plain loads
    if (tid < 32) S0[(tid >> 4) * kImg + nA + nD + (tid & 15)] = 0.f;
    // tile table: (job, mt, nt) of tile i, jobs in order
    if (tid < kFwWaves * kFwTilesPerWave) {
        int i = tid, code = -1;
        for (int j = 0; j < F.J.n; ++j) {
            const int MT = (F.J.j[j].M + 15) >> 4, NT = (F.J.j[j].N + 1 + 15) >> 4;
            if (i < MT * NT) { code = (j << 16) | ((i / NT) << 8) | (i % NT); break; }
            i -= MT * NT;
        }
        TT[tid] = code;
    }
    __syncthreads();
    // this lane's operand addresses of its wave's tiles, once: delta index in bits 0..14, act index in bits 16..30, bit 31: this lane is the bias column
    // (and, wave-uniform, which of the wave's tiles exist / hold their job's bias column: two bit masks -- asking the job table per tile and chunk
    //  costs a scalar load from the kernel arguments each time, more than the tile's four MFMAs)
    unsigned ix[kFwTilesPerWave];
    unsigned long long tiles = 0ull, bias_tiles = 0ull;
#pragma unroll
    for (int q = 0; q < kFwTilesPerWave; ++q) {
        const int code = TT[w + kFwWaves * q];
        ix[q] = 0u;
        if (code >= 0) {
            tiles |= 1ull << q;
            if (16 * (code & 255) + 16 > F.J.j[code >> 16].N) bias_tiles |= 1ull << q;
            const WgradJob& J = F.J.j[code >> 16];
            const int m = 16 * ((code >> 8) & 255) + (lane & 15), n = 16 * (code & 255) + (lane & 15), kk = lane >> 4;
            ix[q] = (unsigned)(nA + kk * LDD + (int)(J.delta - F.del) + m + (m >= J.m_split ? J.m_gap : 0)) | ((unsigned)(kk * LDA + (int)(J.act - F.act) + n) << 16) |
                    (n == J.N ? 0x80000000u : 0u);
        }
    }
    f32x4 acc[kFwTilesPerWave];
#pragma unroll
    for (int q = 0; q < kFwTilesPerWave; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nchunks = (F.K + kFwSamples - 1) / kFwSamples;
    typedef __attribute__((address_space(3))) void lds_v;
    typedef const __attribute__((address_space(1))) void gbl_v;
    // requests of chunk c into image `buf` (asynchronous: vmcnt) -- or, for the last partial chunk / records that are no whole requests, plain copies
    auto request = [&](int c, int buf) {
        float* S = S0 + buf * kImg;
        const int s0 = c * kFwSamples, ns = min(kFwSamples, F.K - s0);
        if (kDma && ns == kFwSamples) {
            const float* ga = F.act + (size_t)s0 * LDA; const float* gd = F.del + (size_t)s0 * LDD;
            for (int u = w; u < nA / 64; u += kFwWaves) __builtin_amdgcn_global_load_lds((gbl_v*)(ga + u * 64 + lane), (lds_v*)(S + u * 64), 4, 0, 0);
            for (int u = w; u < nD / 64; u += kFwWaves) __builtin_amdgcn_global_load_lds((gbl_v*)(gd + u * 64 + lane), (lds_v*)(S + nA + u * 64), 4, 0, 0);
        } else {
            for (int i = tid; i < nA; i += 64 * kFwWaves) S[i] = (i / LDA < ns) ? F.act[(size_t)s0 * LDA + i] : 0.f;
            for (int i = tid; i < nD; i += 64 * kFwWaves) S[nA + i] = (i / LDD < ns) ? F.del[(size_t)s0 * LDD + i] : 0.f;
        }
    };
    int buf = 0;
    if ((int)blockIdx.x < nchunks) request(blockIdx.x, 0);
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's requests of chunk c have landed
        __syncthreads();                                     // everybody's have, and everybody has left the other image (chunk c - stride)
        if (c + (int)gridDim.x < nchunks) request(c + gridDim.x, buf ^ 1);
        const float* S = S0 + buf * kImg;
#pragma unroll
        for (int q = 0; q < kFwTilesPerWave; ++q) {
            if ((tiles >> q) & 1ull) {                       // (wave-uniform)
                const float* dp = S + (ix[q] & 0x7FFFu);
                const float* ap = S + ((ix[q] >> 16) & 0x7FFFu);
                f32x4 a = acc[q];
                if ((bias_tiles >> q) & 1ull) {              // the tile holds the bias column: that lane multiplies by 1
                    const bool ab = (ix[q] >> 31) != 0;
#pragma unroll
                    for (int ks = 0; ks < kFwSamples / 4; ++ks) { const float y = ap[ks * 4 * LDA]; a = mfma16(dp[ks * 4 * LDD], ab ? 1.f : y, a); }
                } else {
#pragma unroll
                    for (int ks = 0; ks < kFwSamples / 4; ++ks) a = mfma16(dp[ks * 4 * LDD], ap[ks * 4 * LDA], a);
                }
                acc[q] = a;
            }
        }
    }
    // partials of this workgroup: the accumulators as they are, one 16-byte store per lane and tile (a store in the Flux layout touches 64 lines
    // per instruction: 15 M four-byte transactions for the GRU's jobs, the larger part of the launch); the reduction maps them to the jobs' outputs
    const int ntiles = fused_tile_count(F.J);
#pragma unroll
    for (int q = 0; q < kFwTilesPerWave; ++q) {
        if ((tiles >> q) & 1ull) ((f32x4*)(F.raw + ((size_t)blockIdx.x * ntiles + (w + kFwWaves * q)) * 256))[lane] = acc[q];
    }
}
// reduction of the accumulator images, level 1: blockIdx.y sums the workgroups [y * per, (y + 1) * per) element by element, in order
__global__ void rnde_latent_reduce_raw_kernel(const float* __restrict__ raw, int G, int per, int E, float* __restrict__ raw2) {
    const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < E / 4; i += gridDim.x * 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int c = g0;
        for (; c + 4 <= g1; c += 4) {
            const f32x4 v0 = ((const f32x4*)(raw + (size_t)c * E))[i], v1 = ((const f32x4*)(raw + (size_t)(c + 1) * E))[i],
                        v2 = ((const f32x4*)(raw + (size_t)(c + 2) * E))[i], v3 = ((const f32x4*)(raw + (size_t)(c + 3) * E))[i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; c < g1; ++c) s += ((const f32x4*)(raw + (size_t)c * E))[i];
        ((f32x4*)(raw2 + (size_t)blockIdx.y * E))[i] = s;
    }
}
// level 2: the segments in order, and the accumulator element (tile, lane, i) goes to its place in its job's output (Flux layout [vec(W); b])
__global__ void rnde_latent_reduce_scatter_kernel(const FusedWgrad F, const float* __restrict__ raw2, int nseg, int E) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256) {
        float s = 0.f;
        for (int c = 0; c < nseg; ++c) s += raw2[(size_t)c * E + e];
        int t = e >> 8; const int lane = (e >> 2) & 63, i = e & 3;
        for (int j = 0; j < F.J.n; ++j) {
            const WgradJob& J = F.J.j[j];
            const int MT = (J.M + 15) >> 4, NT = (J.N + 1 + 15) >> 4;
            if (t < MT * NT) {
                const int m = 16 * (t / NT) + 4 * (lane >> 4) + i, n = 16 * (t % NT) + (lane & 15);
                if (m < J.M && n <= J.N) J.out[(size_t)n * J.M + m] = s;
                break;
            }
            t -= MT * NT;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// rec_to_gen + sampling (time_series.jl:50-59): one thread per (sample, output row); tapes h1 (50) and [mu0; logvar] (40)
// ---------------------------------------------------------------------------------------------------------------------------------
struct EncParams {
    const float* y; const float* p2; const float* eps;      // y: 100 x B; eps: 20 x B
    float* h1; float* out;                                   // tapes: 50 x B, 40 x B
    float* z0; float* mu0; float* logvar; float* kl;         // outputs (20 x B each), kl: per sample
    int B;
};
__global__ __launch_bounds__(64) void rnde_latent_enc_fwd_kernel(const EncParams Q) {
    __shared__ float ys[2 * kL], hs[kRec], os[2 * kLat];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * kL; i += 64) ys[i] = Q.y[(size_t)b * 2 * kL + i];
    __syncthreads();
    const float* W1 = Q.p2; const float* b1 = W1 + 2 * kL * kRec; const float* W2 = b1 + kRec; const float* b2 = W2 + kRec * 2 * kLat;
    if (tid < kRec) {
        float s = b1[tid];
        for (int k = 0; k < 2 * kL; ++k) s = fmaf(W1[k * kRec + tid], ys[k], s);
        s = tanh_f(s);
        hs[tid] = s; Q.h1[(size_t)b * kRec + tid] = s;
    }
    __syncthreads();
    if (tid < 2 * kLat) {
        float s = b2[tid];
        for (int k = 0; k < kRec; ++k) s = fmaf(W2[k * 2 * kLat + tid], hs[k], s);
        os[tid] = s; Q.out[(size_t)b * 2 * kLat + tid] = s;
    }
    __syncthreads();
    if (tid < kLat) {
        const float mu = os[tid], lv = os[kLat + tid];
        Q.mu0[(size_t)b * kLat + tid] = mu; Q.logvar[(size_t)b * kLat + tid] = lv;
        Q.z0[(size_t)b * kLat + tid] = Q.eps[(size_t)b * kLat + tid] * __expf(0.5f * lv) + mu;
        hs[tid] = __expf(lv) + mu * mu - 1.f - lv;
    }
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int i = 0; i < kLat; ++i) s += hs[i]; Q.kl[b] = s / (2.f * kLat); }      // kl_divergence, latent_ode.jl:203-204
}
// reverse: z0-bar (20 x B) and the KL weight -> delta tapes d2 (40 x B: cotangent of [mu0; logvar]), d1 (50 x B: of the tanh layer's
// pre-activation), y-bar (100 x B); the weight gradients are two rnde_latent_gru_wgrad_kernel passes (d2 x h1, d1 x y)
struct EncBwdParams {
    const float* z0b; const float* p2; const float* eps; const float* h1; const float* out;
    float* d2; float* d1; float* yb; float klw; int B;      // klw = lambda_k / B
};
__global__ __launch_bounds__(128) void rnde_latent_enc_bwd_kernel(const EncBwdParams Q) {
    __shared__ float d2s[2 * kLat], d1s[kRec];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* W1 = Q.p2; const float* W2 = W1 + 2 * kL * kRec + kRec;
    if (tid < kLat) {
        const float mu = Q.out[(size_t)b * 2 * kLat + tid], lv = Q.out[(size_t)b * 2 * kLat + kLat + tid], zb = Q.z0b[(size_t)b * kLat + tid];
        const float mub = zb + Q.klw * mu / kLat;
        const float lvb = zb * Q.eps[(size_t)b * kLat + tid] * __expf(0.5f * lv) * 0.5f + Q.klw * (__expf(lv) - 1.f) / (2.f * kLat);
        d2s[tid] = mub; d2s[kLat + tid] = lvb;
        Q.d2[(size_t)b * 2 * kLat + tid] = mub; Q.d2[(size_t)b * 2 * kLat + kLat + tid] = lvb;
    }
    __syncthreads();
    if (tid < kRec) {
        float s = 0.f;
        for (int o = 0; o < 2 * kLat; ++o) s = fmaf(W2[tid * 2 * kLat + o], d2s[o], s);
        const float h = Q.h1[(size_t)b * kRec + tid];
        s *= 1.f - h * h;
        d1s[tid] = s; Q.d1[(size_t)b * kRec + tid] = s;
    }
    __syncthreads();
    if (tid < 2 * kL) {
        float s = 0.f;
        for (int o = 0; o < kRec; ++o) s = fmaf(W1[tid * kRec + o], d1s[o], s);
        Q.yb[(size_t)b * 2 * kL + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// gen_to_data + masked likelihood, forward and reverse in one launch (latent_ode.jl:148, :192-200, :226-236): one workgroup per sample,
// one thread per save time.  ll[b] = sum_{i,t} (-(d^2) / (2 sigma^2) - log sigma - log(2 pi) / 2) / sum(mask): the constants are counted at
// unobserved entries too, as the reference does.  pred-bar goes to the delta tape gD (37 per (t, b)), res-bar = W4^T pred-bar.
// ---------------------------------------------------------------------------------------------------------------------------------
struct DecParams {
    const float* res;      // kLat x T x B
    const float* p4; const float* x;      // x: kNX x T x B (data rows 0..36, mask rows 37..73)
    float* gD;             // [B * T][40]: pred-bar, padded to 40
    float* resb;           // kLat x T x B
    float* ll;             // [B]
    int B, T; float inv_sigma2, c0;      // c0 = -log(sigma) - log(2 pi) / 2
};
__global__ __launch_bounds__(64) void rnde_latent_dec_loss_kernel(const DecParams Q) {
    __shared__ float red[64], Ws[kP4];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < kP4; i += 64) Ws[i] = Q.p4[i];
    const bool tok = t < Q.T;
    const float* xp = Q.x + ((size_t)b * Q.T + (tok ? t : 0)) * kNX;
    float msum = 0.f;
    if (tok) for (int i = 0; i < kIn; ++i) msum += xp[kIn + i];
    red[t] = msum;
    __syncthreads();
    float M = 0.f;
    for (int i = 0; i < 64; ++i) M += red[i];      // (fixed order: the same on every thread)
    __syncthreads();
    float lsum = 0.f;
    if (tok) {
        float z[kLat], rb[kLat];
        const float* rp = Q.res + ((size_t)b * Q.T + t) * kLat;
#pragma unroll
        for (int j = 0; j < kLat; ++j) { z[j] = rp[j]; rb[j] = 0.f; }
        float* gp = Q.gD + ((size_t)b * Q.T + t) * 40;
        const float gscale = Q.inv_sigma2 / (M * (float)Q.B);
        for (int i = 0; i < kIn; ++i) {
            float pr = Ws[kLat * kIn + i];
#pragma unroll
            for (int j = 0; j < kLat; ++j) pr = fmaf(Ws[j * kIn + i], z[j], pr);
            const float mk = xp[kIn + i];
            const float d = pr * mk - xp[i] * mk;
            lsum += -d * d * 0.5f * Q.inv_sigma2 + Q.c0;
            const float g = d * mk * gscale;
            gp[i] = g;
#pragma unroll
            for (int j = 0; j < kLat; ++j) rb[j] = fmaf(Ws[j * kIn + i], g, rb[j]);
        }
        gp[37] = 0.f; gp[38] = 0.f; gp[39] = 0.f;
        float* ob = Q.resb + ((size_t)b * Q.T + t) * kLat;
#pragma unroll
        for (int j = 0; j < kLat; ++j) ob[j] = rb[j];
    }
    red[t] = lsum;
    __syncthreads();
    if (t == 0) { float s = 0.f; for (int i = 0; i < 64; ++i) s += red[i]; Q.ll[b] = s / M; }
}
// nll = -mean(ll), kl = mean(kl): two floats (256 threads, per-thread partial sums in index order, then a fixed tree: deterministic)
__global__ __launch_bounds__(256) void rnde_latent_loss_kernel(const float* __restrict__ ll, const float* __restrict__ kl, int B, float* __restrict__ out) {
    __shared__ double sa[256], sc[256];
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < B; i += 256) { a += ll[i]; c += kl[i]; }
    sa[threadIdx.x] = a; sc[threadIdx.x] = c;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sa[threadIdx.x] += sa[threadIdx.x + st]; sc[threadIdx.x] += sc[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(-sa[0] / B); out[1] = (float)(sc[0] / B); }
}

// Flux.Optimise.Optimiser(InvDecay(gamma), AdaMax(eta, (beta1, beta2))) on one flat group, in place (latent_ode.jl:108; Flux 0.11 `apply!`):
// g <- g / (1 + gamma n); m <- beta1 m + (1 - beta1) g; u <- max(beta2 u, |g|); p <- p - eta / (1 - beta1_pow) * m / (u + eps)
__global__ void rnde_adamax_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ u, long long len,
                                   float inv_decay, float eta_hat, float beta1, float beta2, float eps) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < len; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * inv_decay;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float ui = fmaxf(beta2 * u[i], fabsf(gi));
        m[i] = mi; u[i] = ui;
        p[i] -= eta_hat * mi / (ui + eps);
    }
}

}  // namespace rnde_lat
