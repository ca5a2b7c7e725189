// rnde_stage.h -- "stage engine": weight-stationary forward kernels, one launch per Runge-Kutta stage.
//
// Why a second engine: the column-owner kernel re-streams all weights per workgroup and
// occupies only B/8 CUs; at the reference's batch (512) that is 64 CUs and a ~60 us/attempt floor.  Here the
// work of ONE f evaluation is spread over R row-blocks x C column-tiles (7 x 32 = 224 workgroups for MNIST):
//
//   phase A  h      = tanh( sum over row-blocks of z1-partial slabs + w1t*t + b1 )      [H x 16]   (redundant per row-block)
//   phase B  k_s    = act2( W2ext[rows of this block, :] * [h; t; 1] )                   [16*WT rows x 16 cols]
//   phase C  store k_s; g_{s+1} = uprev + dt * sum_j a_{s+1,j} k_j  (this block's rows)  (last stage: error partial)
//   phase D  z1-partial slab = W1[:, rows of this block] * g_{s+1}[rows]                 [H x 16]  -> next launch
//
// so each workgroup touches only its own 1/R slice of both weight matrices (~90 KB instead of 740 KB), the
// Dense layers run on v_mfma_f32_16x16x4_f32 (4x the MACs per operand register of the 4x4x1 form), and the
// split-K reduction of layer 1 crosses the kernel boundary as a slab (deterministic, no atomics).
// The controller prologue, the tape layout, StepState/StepMeta and the host loop are shared with the
// column-owner engine (rnde_fwd.h).
#pragma once
#include "rnde_fwd.h"

namespace rnde {

constexpr int kSCB = 16;      // batch columns per workgroup
constexpr int kSMaxW = 8;     // max waves (= row tiles per row block) per workgroup
constexpr int kSMaxHT = 8;    // max 16-row tiles of the hidden layer (H + 1 <= 128)

struct StageParams {
    StepParams F;             // shared fields (x, f0, arena, ctl, meta, errpart, initpart, dims, tolerances ...)
    const float* p;           // flat parameters (for w1t, b1)
    const f32x4* pwB;         // [MT][K2b][64]   layer 2 rows      (forward)  /  W1x^T rows (reverse)
    const f32x4* pwD;         // [HT][MT][64]    layer 1 K-slices  (forward)  /  W2x^T K-slices (reverse)
    float* slab;              // [2][C][R][HT][64][4]
    int MT, WT, R, C, HT, K2b;  // row tiles, waves per block, row blocks, column tiles, hidden tiles, k16 blocks of layer 2
    int Bpad16;
};

// position of k inside the permuted LDS B-operand image: lane (kk = l>>4) reads 4 consecutive floats that
// feed 4 successive 16x16x4 MFMAs (k = 16*kb + 4*q + kk for q = 0..3)
__device__ __forceinline__ int kperm(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// acc + d * d, multiply and add rounded separately.  Every kernel of the stage engine forms the norm partials (error estimate, stiffness
// estimate) with this: left to the compiler, contraction depends on the surrounding code (a row condition, the vectoriser), and kernels
// that must agree bit for bit (7-launch, one-launch attempt, two-tile, wide, one-launch solve) differed in the last bit of eigen_est.
__device__ __forceinline__ float add_square_unfused(float acc, float d) {
#pragma clang fp contract(off)
    const float m = d * d;
    return acc + m;
}
// explicit fused multiply-adds for the stage combinations: rnde_stage_kernel and rnde_stage_attempt_kernel must round
// identically (their outputs are compared bit for bit), so contraction is not left to the compiler
__device__ __forceinline__ f32x4 fma4(float s, f32x4 a, f32x4 c) { return __builtin_elementwise_fma((f32x4){s, s, s, s}, a, c); }

// packed A operands for the stage engine: element (tile T, block kb, lane l, q) = Wsel[16*T + (l&15)][16*kb + 4*q + (l>>4)]
// which: 0 pwB fwd  = W2ext (rows D, k <= H: W2 incl. time col; k == H+1: b2)
//        1 pwD fwd  = W1x   (rows H, k < D)
//        2 pwB bwd  = W1x^T (rows D, k < H):   W1[k][row]
//        3 pwD bwd  = W2xt^T (rows H+1, k < D): row m < H: W2[k][m]; m == H: W2[k][H] (time column)
__device__ __forceinline__ f32x4 stage_pack_elem(const float* __restrict__ p, int which, int D, int H, int Kb, long long i) {
    const float* W1 = p;
    const float* b1 = W1 + (size_t)H * (D + 1);
    const float* W2 = b1 + H;
    const float* b2 = W2 + (size_t)D * (H + 1);
    const int l = (int)(i & 63);
    const int kb = (int)((i >> 6) % Kb);
    const int T = (int)((i >> 6) / Kb);
    const int m = 16 * T + (l & 15);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 16 * kb + 4 * q + (l >> 4);
        float w = 0.f;
        if (which == 0) { if (m < D) w = k <= H ? W2[(size_t)k * D + m] : (k == H + 1 ? b2[m] : 0.f); }
        else if (which == 1) { if (m < H && k < D) w = W1[(size_t)k * H + m]; }
        else if (which == 2) { if (m < D && k < H) w = W1[(size_t)m * H + k]; }
        else { if (m <= H && k < D) w = W2[(size_t)m * D + k]; }
        v[q] = w;
    }
    return v;
}
static __global__ void rnde_stage_pack_kernel(const float* __restrict__ p, f32x4* __restrict__ dst, int which, int D, int H,
                                       int MTrows, int Kb) {
    const long long total = (long long)MTrows * Kb * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dst[i] = stage_pack_elem(p, which, D, H, Kb, i);
}

// Everything a taped solve derives from the parameter vector, in ONE launch (was six pack launches and a copy, ~5 us each, in front
// of every training step): blockIdx.y selects the job -- a stage-engine pack (kind 0), a column-owner pack for the kernels of the
// initial-step rule (kind 1), or the tape's own copy of p (kind 2).
struct PackJob { void* dst; const float* src; long long total; int kind, which, kdim, pad; };   // src: kind 2 only (NULL = the parameter vector)
struct PackJobs { PackJob j[8]; };
static __global__ void rnde_pack_all_kernel(const float* __restrict__ p, const PackJobs J, int D, int H) {
    const PackJob job = J.j[blockIdx.y];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < job.total; i += (long long)gridDim.x * blockDim.x) {
        if (job.kind == 0) ((f32x4*)job.dst)[i] = stage_pack_elem(p, job.which, D, H, job.kdim, i);
        else if (job.kind == 2) ((float*)job.dst)[i] = (job.src ? job.src : p)[i];
        else ((f32x4*)job.dst)[i] = ((const f32x4*)(job.src ? job.src : p))[i];      // kind 3: 16-byte copy (total counts float4s)
    }
}

// Dense output of the attempt that has just been accepted (record Rp: uprev/k1 copies, k2..k7, unew), for save indices
// [lo, hi): 4 rows of one column per call.  A save time equal to the new t copies unew (as the reference does).
__device__ __forceinline__ void dense_points(const StepParams& P, const RecLayout& L, const float* Rp, float tp, float dtp_, float tnew,
                                             int lo, int hi, size_t co, int gcol, int r0, bool colok, bool vec) {
    const f32x4 up = ld_tile(Rp + L.upc() + co, r0, P.D, true, vec);
    f32x4 k[7];
    k[0] = ld_tile(Rp + L.k1c() + co, r0, P.D, true, vec);
#pragma unroll
    for (int j = 1; j < 7; ++j) k[j] = ld_tile(Rp + L.k(j + 1) + co, r0, P.D, true, vec);
    const f32x4 un = ld_tile(Rp + L.unew() + co, r0, P.D, true, vec);
    for (int idx = lo; idx < hi; ++idx) {
        const float ts = P.sv_t[idx];
        f32x4 o = un;
        if (ts != tnew) {
            float b[7];
            dense_weights((ts - tp) / dtp_, b);
            f32x4 acc = b[0] * k[0];
#pragma unroll
            for (int j = 1; j < 7; ++j) acc += b[j] * k[j];
            o = up + dtp_ * acc;
        }
        st_tile(P.sv_out + ((size_t)gcol * P.nsave + idx) * P.D, r0, P.D, colok, vec && ((P.D & 3) == 0), o);
    }
}

enum { SM_START = 0, SM_STAGE = 1, SM_LAST = 2, SM_I1 = 3, SM_I2 = 4, SM_I3 = 5, SM_I4 = 6, SM_FEVAL1 = 7, SM_FEVAL2 = 8 };

// 4 consecutive rows of one column (16x16 D-fragment ownership: col = lane & 15, rows 4*(lane>>4) + reg)
__device__ __forceinline__ f32x4 ld4(const float* colbase, int r0, int D, bool ok, bool vec) { return ld_tile(colbase, r0, D, ok, vec); }
__device__ __forceinline__ void st4(float* colbase, int r0, int D, bool ok, bool vec, f32x4 v) { st_tile(colbase, r0, D, ok, vec, v); }

template <int ACT2, int MODE>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_stage_kernel(const StageParams Q, const int n, const int s) {
    const StepParams& P = Q.F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KH = 16 * Q.K2b + 4, KG = 16 * Q.WT + 4;   // LDS column strides of the two B-operand images
    float* HL = smem;                    // [16][KH]  hidden activations (+ t, 1), permuted k
    float* GL = HL + kSCB * KH;          // [16][KG]  this block's rows of the stage input, permuted k
    float* RED = GL + kSCB * KG;         // [32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = blockIdx.x / Q.C, ct = blockIdx.x - rb * Q.C;   // all row blocks of a column tile share blockIdx % 8 when C % 8 == 0
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (P.D & 3) == 0;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    const int T = rb * Q.WT + w;                       // this wave's row tile
    const int r0 = 16 * T + 4 * (lane >> 4);           // first of its 4 rows
    const bool tile_ok = T < Q.MT;
    const RecLayout L{(long long)P.D * P.Bpad, (long long)P.H * P.Bpad};
    constexpr bool kHasA = (MODE == SM_STAGE || MODE == SM_LAST || MODE == SM_I2 || MODE == SM_I4 || MODE == SM_FEVAL2);
    constexpr bool kHasD = (MODE == SM_START || MODE == SM_STAGE || MODE == SM_I1 || MODE == SM_I3 || MODE == SM_FEVAL1);

#ifdef RNDE_DIAG
#define SSTAMP(i) do { if (MODE == SM_STAGE && s == 3 && P.dbg_out && blockIdx.x == 0 && lane == 0) ((unsigned long long*)P.dbg_out)[(w * 8 + (i))] = clock64(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    SSTAMP(0);
    // After a kernel boundary every read is served from Infinity Cache / HBM (the per-XCD L2s are written back
    // and invalidated), ~2 us per DEPENDENT round trip: issue every independent load of the launch up front --
    // controller state, weights, slabs -- and the state-array operands as soon as the controller state is in.
    StepState S0;
    if constexpr (MODE == SM_STAGE || MODE == SM_LAST) S0 = P.ctl[n & 1];
    f32x4 c_up = {0.f, 0.f, 0.f, 0.f}, c_un = {0.f, 0.f, 0.f, 0.f}, c_k[6];
    bool c_early = false;
    if constexpr (MODE == SM_STAGE || MODE == SM_LAST) {
        if (P.tape && tile_ok) {   // taped forward: this attempt's record is record n, (uprev, k1) were copied into it by SM_START
            const float* Rn = P.arena + (long long)n * P.rec_stride;
            const size_t co = (size_t)gcol * P.D;
            c_up = ld4(Rn + L.upc() + co, r0, P.D, true, vec);
            c_k[0] = ld4(Rn + L.k1c() + co, r0, P.D, true, vec);
#pragma unroll
            for (int j = 1; j < 6; ++j) if (j < s) c_k[j] = ld4(Rn + L.k(j + 1) + co, r0, P.D, true, vec);
            if constexpr (MODE == SM_LAST) c_un = ld4(Rn + L.unew() + co, r0, P.D, true, vec);
            c_early = true;
        }
    }
    // ---- weights: this workgroup's 1/R slice, ~90 KB ----
    f32x4 wB[kSMaxHT];   // phase B A-operands: row tile T, all K2b blocks   (K2b <= 8)
    f32x4 wD[kSMaxW];    // phase D A-operands: hidden tile(s) of this wave, the block's WT k16 blocks (first hidden tile)
    if constexpr (kHasA) {
#pragma unroll
        for (int kb = 0; kb < kSMaxHT; ++kb)
            if (kb < Q.K2b && tile_ok) wB[kb] = Q.pwB[((size_t)T * Q.K2b + kb) * 64 + lane];
    }
    if constexpr (kHasD) {
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb)
            if (kb < Q.WT && w < Q.HT && rb * Q.WT + kb < Q.MT) wD[kb] = Q.pwD[((size_t)w * Q.MT + rb * Q.WT + kb) * 64 + lane];
    }

    // z1 slabs of the previous launch (independent of the controller state: issue now)
    f32x4 zs = {0.f, 0.f, 0.f, 0.f};
    f32x4 zr[kSMaxW];
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kHasA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hr = 16 * w + 4 * (lane >> 4) + i;
            if (hr < P.H) { w1t_own[i] = Q.p[(size_t)P.H * P.D + hr]; b1_own[i] = Q.p[(size_t)P.H * (P.D + 1) + hr]; }
        }
        const int par0 = (MODE == SM_STAGE || MODE == SM_LAST) ? (s & 1) : 0;
        const f32x4* sl0 = (const f32x4*)Q.slab + (((size_t)par0 * Q.C + ct) * Q.R) * Q.HT * 64;
        if (w < Q.HT) {
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zr[r] = sl0[((size_t)r * Q.HT + w) * 64 + lane];
        }
    }

    // ---- scalars of this attempt ----
    float t = P.t0, dt = 0.f;
    int live = -1, rec = 0;
    if constexpr (MODE == SM_START) {
        const StepState S = advance_state(P, n, lane, writer, &P.ctl[n & 1]);
        if (P.nsave > 0) {
            // saveat (reference neural_ode.jl:79-108: the {R,true} methods): points inside the step just accepted
            const int lo = (n == 0) ? 0 : P.ctl[(n - 1) & 1].next_save, hi = S.next_save;
            if (hi > lo && tile_ok) {
                if (n == 0) {
                    st_tile(P.sv_out + (size_t)gcol * P.nsave * P.D, r0, P.D, colok, vec,
                            ld_tile(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0));
                } else {
                    const StepState pv = P.ctl[(n - 1) & 1];
                    const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                    const float* Rp = P.arena + (long long)S.live * P.rec_stride;     // accepted => it is the live record
                    dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, (size_t)gcol * P.D, gcol, r0, colok, vec);
                }
            }
        }
        if (S.done) return;
        t = S.t; dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp; live = S.live;
    } else if constexpr (MODE == SM_STAGE || MODE == SM_LAST) {
        const StepState S = S0;
        if (S.done) return;
        t = S.t; dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp; live = S.live;
    }
    if constexpr (MODE == SM_START || MODE == SM_STAGE || MODE == SM_LAST) rec = P.tape ? n + P.rec_shift : (live == 0 ? 1 : 0);
    float* R = P.arena + (long long)rec * P.rec_stride;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (live >= 0) { const float* Rl = P.arena + (long long)live * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }

    // slab sum first (these loads were issued together with the controller state, so they have landed); only then
    // issue the state-array loads -- a wait placed after them would have to cover them too (conditional loads
    // make the compiler fall back to vmcnt(0))
    if constexpr (kHasA) {
        if (w < Q.HT) {
            const int par0 = (MODE == SM_STAGE || MODE == SM_LAST) ? (s & 1) : 0;
            const f32x4* sl0 = (const f32x4*)Q.slab + (((size_t)par0 * Q.C + ct) * Q.R) * Q.HT * 64;
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zs += zr[r];   // fixed order: deterministic
            for (int r = kSMaxW; r < Q.R; ++r) zs += sl0[((size_t)r * Q.HT + w) * 64 + lane];
        }
    }
    // phase-C operands (state arrays): issue now so they land under phases A and B
    if constexpr (MODE == SM_START || MODE == SM_STAGE || MODE == SM_LAST) {
        if (tile_ok && !c_early) {
            c_up = ld4(upsrc + (size_t)gcol * P.D, r0, P.D, upok, upvec);
            c_k[0] = ld4(k1p + (size_t)gcol * P.D, r0, P.D, true, vec);
            if constexpr (MODE != SM_START) {
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) c_k[j] = ld4(R + L.k(j + 1) + (size_t)gcol * P.D, r0, P.D, true, vec);
            }
            if constexpr (MODE == SM_LAST) c_un = ld4(R + L.unew() + (size_t)gcol * P.D, r0, P.D, true, vec);
        }
    }

    float dt0 = 0.f;
    if constexpr (MODE == SM_I3) {   // initial-step heuristic: dt0 from the norms of u0 and f0
        const double N = (double)P.D * (double)P.Bn;
        const double s0 = sum_partials(P.initpart, P.nwg, lane), s1 = sum_partials(P.initpart + P.nwg, P.nwg, lane);
        const float d0 = (float)sqrt(s0 / N), d1 = (float)sqrt(s1 / N), dtmax = P.t1 - P.t0;
        int c0 = 0, cl = 0;
        if (d0 < 1e-5f || d1 < 1e-5f) { dt0 = 1e-6f; c0 = 1; } else dt0 = (d0 / d1) / 100.f;
        if (dtmax < dt0) { dt0 = dtmax; cl = 1; }
        if (writer) { P.initrec->d0 = d0; P.initrec->d1 = d1; P.initrec->dt0 = dt0; P.initrec->dt0_const = c0; P.initrec->dt0_clamped = cl; }
    }
    if constexpr (MODE == SM_I4) dt0 = P.initrec->dt0;

    // stage time and tape slots
    float ts = t;
    float* hdst = nullptr; float* kdst = nullptr;
    if constexpr (MODE == SM_STAGE || MODE == SM_LAST) { ts = fmaf(kTsC[s], dt, t); hdst = R + L.h(s + 1); kdst = R + L.k(s + 1); }
    if constexpr (MODE == SM_I2) { ts = P.t0; hdst = P.h0; kdst = P.f0; }
    if constexpr (MODE == SM_I4) { ts = P.t0 + dt0; hdst = P.h1; kdst = P.f1; }
    if constexpr (MODE == SM_FEVAL2) { ts = P.forced_t; }

    f32x4 kv = {0.f, 0.f, 0.f, 0.f};
    SSTAMP(1);
    if constexpr (kHasA) {

        // ---- phase A: hidden activations of this column tile from the z1 slabs of the previous launch ----
        const float* W1t = Q.p + (size_t)P.H * P.D;           // time column of W1 (H x (D+1), column-major)
        const float* b1 = Q.p + (size_t)P.H * (P.D + 1);
        const int par = (MODE == SM_STAGE || MODE == SM_LAST) ? (s & 1) : 0;
        const f32x4* sl = (const f32x4*)Q.slab + (((size_t)par * Q.C + ct) * Q.R) * Q.HT * 64;
        for (int ht = w; ht < Q.HT; ht += Q.WT) {
            f32x4 z = zs;
            if (ht != w) {
                z = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < Q.R; ++r) z += sl[((size_t)r * Q.HT + ht) * 64 + lane];
            }
            const int h0 = 16 * ht + 4 * (lane >> 4);
            // two tanh per instruction (v_pk_fma_f32): the 4 rows of this lane as two pairs
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                pre[i] = (hr < P.H) ? fmaf((ht == w) ? w1t_own[i] : W1t[hr], ts, z[i]) + ((ht == w) ? b1_own[i] : b1[hr]) : 0.f;
            }
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            const float th4[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                float v = 0.f;
                if (hr < P.H) {
                    v = th4[i];
                    if (rb == 0 && hdst) hdst[(size_t)gcol * P.H + hr] = v;
                } else if (hr == P.H) v = ts;
                else if (hr == P.H + 1) v = 1.f;
                if (hr < 16 * Q.K2b) HL[col * KH + kperm(hr)] = v;
            }
        }
        // rows beyond the reduced tiles (when H + 2 spills into a further 16-block)
        if (Q.K2b > Q.HT) {
            for (int i = tid; i < kSCB * 16 * Q.K2b; i += blockDim.x) {
                const int c = i / (16 * Q.K2b), k = i - c * 16 * Q.K2b;
                if (k >= 16 * Q.HT) HL[c * KH + kperm(k)] = (k == P.H) ? ts : (k == P.H + 1 ? 1.f : 0.f);
            }
        }
        SSTAMP(2);
        __syncthreads();
        SSTAMP(3);
        // ---- phase B: this wave's 16 rows of layer 2 ----
        if (tile_ok) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* hb = HL + col * KH + 4 * (lane >> 4);
            f32x4 bf[kSMaxHT];
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) if (kb < Q.K2b) bf[kb] = *(const f32x4*)(hb + 16 * kb);
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) {
                if (kb < Q.K2b) {
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
            }
            kv = acc0 + acc1;
            if (ACT2) {
                const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) kv[i] = (r0 + i < P.D) ? kv[i] : 0.f;
        }
    }

    SSTAMP(4);
    // ---- phase C: element-wise work on this wave's 16 rows x 16 columns ----
    f32x4 v = {0.f, 0.f, 0.f, 0.f};   // this wave's rows of the next layer-1 input
    float part0 = 0.f, part1 = 0.f, part2 = 0.f;
    if (tile_ok) {
        if constexpr (MODE == SM_START) {
            v = fma4(dt, kFwdShift[0][0] * c_k[0], c_up);
            if (P.tape) st4(R + L.g(2) + (size_t)gcol * P.D, r0, P.D, true, vec, v);
            if (P.tape || P.nsave > 0) {
                st4(R + L.upc() + (size_t)gcol * P.D, r0, P.D, true, vec, c_up);
                st4(R + L.k1c() + (size_t)gcol * P.D, r0, P.D, true, vec, c_k[0]);
            }
        } else if constexpr (MODE == SM_STAGE) {
            st4(kdst + (size_t)gcol * P.D, r0, P.D, true, vec, kv);
            f32x4 acc = tsA_rt(s + 1, 0) * c_k[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(tsA_rt(s + 1, j), c_k[j], acc);
            acc = fma4(tsA_rt(s + 1, s), kv, acc);
            v = fma4(dt, acc, c_up);
            if (s == 5) st4(R + L.unew() + (size_t)gcol * P.D, r0, P.D, true, vec, v);
            else if (P.tape) st4(R + L.g(s + 2) + (size_t)gcol * P.D, r0, P.D, true, vec, v);
        } else if constexpr (MODE == SM_LAST) {
            st4(kdst + (size_t)gcol * P.D, r0, P.D, true, vec, kv);
            const f32x4 up = c_up, un = c_un;
            f32x4 acc = kTsBt[0] * c_k[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = fma4(kTsBt[j], c_k[j], acc);
            acc = fma4(kTsBt[6], kv, acc);
            if (colok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ut = dt * acc[i];
                    const float sk = P.abstol + fmaxf(fabsf(up[i]), fabsf(un[i])) * P.reltol;
                    const float r = ut / sk;
                    part0 = add_square_unfused(part0, r);
                }
                if (P.reg_kind >= 2) {   // stiffness estimate partials: ||k7 - k6||^2, ||unew - g6||^2
                    f32x4 g6 = tsA_rt(5, 0) * c_k[0];
#pragma unroll
                    for (int j = 1; j < 5; ++j) g6 = fma4(tsA_rt(5, j), c_k[j], g6);
                    g6 = fma4(dt, g6, up);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (r0 + i < P.D) {
                            const float d1 = kv[i] - c_k[5][i], d2 = un[i] - g6[i];
                            part1 = add_square_unfused(part1, d1); part2 = add_square_unfused(part2, d2);
                        }
                    }
                }
            }
        } else if constexpr (MODE == SM_I1 || MODE == SM_FEVAL1) {
            v = ld4(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0);
        } else if constexpr (MODE == SM_I2) {
            st4(kdst + (size_t)gcol * P.D, r0, P.D, true, vec, kv);
            const f32x4 xv = ld4(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0);
            if (colok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (r0 + i < P.D) {
                        const float sk = P.abstol + fabsf(xv[i]) * P.reltol;
                        const float a = xv[i] / sk, b = kv[i] / sk;
                        part0 += a * a; part1 += b * b;
                    }
                }
            }
        } else if constexpr (MODE == SM_I3) {
            const f32x4 xv = ld4(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0);
            const f32x4 f0 = ld4(P.f0 + (size_t)gcol * P.D, r0, P.D, true, vec);
            v = xv + dt0 * f0;
            st4(P.u1 + (size_t)gcol * P.D, r0, P.D, true, vec, v);
        } else if constexpr (MODE == SM_I4) {
            st4(kdst + (size_t)gcol * P.D, r0, P.D, true, vec, kv);
            const f32x4 xv = ld4(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0);
            const f32x4 f0 = ld4(P.f0 + (size_t)gcol * P.D, r0, P.D, true, vec);
            if (colok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (r0 + i < P.D) {
                        const float sk = P.abstol + fabsf(xv[i]) * P.reltol;
                        const float a = (kv[i] - f0[i]) / sk;
                        part0 += a * a;
                    }
                }
            }
        } else if constexpr (MODE == SM_FEVAL2) {
            st4(P.dbg_out + (size_t)gcol * P.D, r0, P.D, colok, false, kv);
        }
    }

    if constexpr (kHasD) {
        // ---- phase D: layer-1 partial pre-activations of this row block -> slab for the next launch ----
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[col * KG + kperm(16 * w + 4 * (lane >> 4) + i)] = (tile_ok && r0 + i < P.D) ? v[i] : 0.f;
        SSTAMP(5);
        __syncthreads();
        SSTAMP(6);
        const int par = (MODE == SM_START) ? 1 : (MODE == SM_STAGE ? ((s + 1) & 1) : 0);
        f32x4* sl = (f32x4*)Q.slab + ((((size_t)par * Q.C + ct) * Q.R + rb) * Q.HT) * 64;
        const float* gbp = GL + col * KG + 4 * (lane >> 4);
        f32x4 bg[kSMaxW];
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb) if (kb < Q.WT) bg[kb] = *(const f32x4*)(gbp + 16 * kb);
        if (w < Q.HT) {   // this wave's own hidden tile: A operands were prefetched at kernel entry
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < Q.WT && rb * Q.WT + kb < Q.MT) {
                    acc0 = mfma16(wD[kb][0], bg[kb][0], acc0);
                    acc1 = mfma16(wD[kb][1], bg[kb][1], acc1);
                    acc0 = mfma16(wD[kb][2], bg[kb][2], acc0);
                    acc1 = mfma16(wD[kb][3], bg[kb][3], acc1);
                }
            }
            sl[(size_t)w * 64 + lane] = acc0 + acc1;
        }
        for (int ht = w + Q.WT; ht < Q.HT; ht += Q.WT) {   // only when HT > WT (tiny state dimension)
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < Q.WT && rb * Q.WT + kb < Q.MT) {
                    const f32x4 a = Q.pwD[((size_t)ht * Q.MT + rb * Q.WT + kb) * 64 + lane];
                    acc0 = mfma16(a[0], bg[kb][0], acc0);
                    acc1 = mfma16(a[1], bg[kb][1], acc1);
                    acc0 = mfma16(a[2], bg[kb][2], acc0);
                    acc1 = mfma16(a[3], bg[kb][3], acc1);
                }
            }
            sl[(size_t)ht * 64 + lane] = acc0 + acc1;
        }
        SSTAMP(7);
    }

    if constexpr (MODE == SM_LAST || MODE == SM_I2 || MODE == SM_I4) {
        part0 = wave_sum_f(part0); part1 = wave_sum_f(part1); part2 = wave_sum_f(part2);
        if (lane == 0) { RED[w] = part0; RED[8 + w] = part1; RED[16 + w] = part2; }
        __syncthreads();
        if (tid == 0) {
            float sa = 0.f, sb = 0.f, sc = 0.f;
            for (int i = 0; i < Q.WT; ++i) { sa += RED[i]; sb += RED[8 + i]; sc += RED[16 + i]; }
            if constexpr (MODE == SM_LAST) {
                float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
                ep[blockIdx.x] = sa; ep[P.nwg + blockIdx.x] = sb; ep[2 * P.nwg + blockIdx.x] = sc;
            }
            else if constexpr (MODE == SM_I2) { P.initpart[blockIdx.x] = sa; P.initpart[P.nwg + blockIdx.x] = sb; }
            else P.initpart[2 * P.nwg + blockIdx.x] = sa;
        }
    }
}

// finish kernel for the stage engine: final controller update + copy-out
static __global__ __launch_bounds__(256) void rnde_stage_finish_kernel(const StageParams Q, const int n, float* __restrict__ u_out) {
    const StepParams& P = Q.F;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    // n < 0: the final state is already in P.ctl_final (the one-launch solve, rnde_stage_solve.h, leaves it there)
    const StepState S = n < 0 ? *P.ctl_final : advance_state(P, n, lane, writer, P.ctl_final);
    const RecLayout L{(long long)P.D * P.Bpad, (long long)P.H * P.Bpad};
    if (P.nsave > 0 && n > 0) {
        const StepState pv = P.ctl[(n - 1) & 1];
        const int lo = pv.next_save, hi = S.next_save;
        if (hi > lo && !pv.done) {     // the last launched attempt was accepted and covers save times
            const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
            const float* Rp = P.arena + (long long)S.live * P.rec_stride;
            const bool vec = (P.D & 3) == 0;
            const int ntile = (P.D + 3) / 4;
            for (long long i = blockIdx.x * 256LL + tid; i < (long long)ntile * P.B; i += (long long)gridDim.x * 256) {
                const int gcol = (int)(i / ntile), r0 = 4 * (int)(i % ntile);
                dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, (size_t)gcol * P.D, gcol, r0, true, vec);
            }
        }
    }
    if (!u_out) return;
    const float* src = S.live < 0 ? P.x : P.arena + (long long)S.live * P.rec_stride + L.unew();
    const long long total = (long long)P.D * P.B;
    if (((total & 3) == 0) && (((reinterpret_cast<size_t>(u_out) | reinterpret_cast<size_t>(src)) & 15) == 0)) {
        for (long long i = blockIdx.x * 256LL + tid; i < total / 4; i += (long long)gridDim.x * 256) ((f32x4*)u_out)[i] = ((const f32x4*)src)[i];
    } else {
        for (long long i = blockIdx.x * 256LL + tid; i < total; i += (long long)gridDim.x * 256) u_out[i] = src[i];
    }
}

}  // namespace rnde
