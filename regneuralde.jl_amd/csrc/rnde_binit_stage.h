// Reverse of the initial-step rule (k1 = f(u0, t0), u1 = u0 + dt0 k1, f1 = f(u1, t0 + dt0), the three norms) on the STAGE engine:
// weight-stationary workgroups (row block rb x column tile ct, one 16-row tile per wave) as rnde_bstage_kernel, four launches
//   STEP 0   scalars of phase 1, z2bar of f1, phase D (hbar partial -> slab, parity 0)
//   STEP 1   phase A (hbar, z1bar of f1), phase B (u1-bar), partial sums {<u1-bar, f0>, tau}
//   STEP 2   scalars of phase 2 (sums of those partials), z2bar of f0 and the direct part of x-bar, phase D (parity 1)
//   STEP 3   phase A, phase B, x-bar += u0-bar of f0, partial sum {tau}
// in place of the two column-owner launches (rnde_binit_kernel, 2 x ~32 us on 64 workgroups that stream the weights through L2 and triple that
// the moment anything else streams beside them).  Same arithmetic per element (rnde_bwd.h: rnde_binit_kernel / f_bwd); the sums over workgroups
// run over the stage engine's R x C partials instead of the column owner's B / 8 -- a different association of the same terms.
#pragma once
#include "rnde_bstage.h"

namespace rnde {

template <int ACT2, int STEP>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_binit_stage_kernel(const BStageParams Q) {
#pragma clang fp contract(off)
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KZ = 16 * Q.KHb + 4, KG = 16 * Q.WT + 4;
    float* ZL = smem;                    // [16][KZ]  z1bar (K = hidden), permuted k
    float* GL = ZL + kSCB * KZ;          // [16][KG]  this block's rows of z2bar, permuted k
    float* RED = GL + kSCB * KG;         // [32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = blockIdx.x / Q.C, ct = blockIdx.x - rb * Q.C;
    const int wg = blockIdx.x;
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (P.D & 3) == 0;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    const int T = rb * Q.WT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const bool tile_ok = T < Q.MT;
    const long long A = (long long)P.D * P.Bpad, HB = (long long)P.H * P.Bpad;
    const size_t co = (size_t)gcol * P.D;
    const double N = (double)P.D * (double)P.Bn;
    const InitRec ir = *P.initrec;
    const float dt0 = ir.dt0;
    constexpr bool kD = (STEP == 0 || STEP == 2);      // the launches that end in phase D
    constexpr int kPhase = STEP < 2 ? 1 : 2;
    constexpr int par = STEP < 2 ? 0 : 1;              // slab parity of the phase

    float dot = 0.f, tau = 0.f;
    if constexpr (kD) {
        f32x4 wD[kSMaxW];
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb)
            if (kb < Q.WT && w < Q.HT && rb * Q.WT + kb < Q.MT) wD[kb] = Q.pwDt[((size_t)w * Q.MT + rb * Q.WT + kb) * 64 + lane];
        f32x4 xv = {0.f, 0.f, 0.f, 0.f}, f0v = xv, f1v = xv, ub1 = xv, Uv = xv, K1v = xv;
        if (tile_ok) {
            xv = ld4(P.x + co, r0, P.D, colok, P.xvec != 0);
            f0v = ld4(P.f0 + co, r0, P.D, true, vec);
            f1v = ld4(P.f1 + co, r0, P.D, true, vec);
            if constexpr (kPhase == 2) {
                ub1 = ld4(Bq.UB1 + co, r0, P.D, true, vec);
                Uv = ld4(Bq.U + co, r0, P.D, true, vec);
                K1v = ld4(Bq.K1 + co, r0, P.D, true, vec);
            }
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};                 // z2bar of this lane's four rows
        if constexpr (kPhase == 1) {
            // ---- scalars of phase 1 (rnde_binit_kernel, PHASE == 1), identical in every wave ----
            double tb, dtpb, qoldb, t1b, t0b;
            finish_attempt_scalars(Bq, 0, lane, tb, dtpb, qoldb, t1b, t0b);
            const double dtb = Bq.track_initdt ? dtpb : 0.0;
            double dt0b = 0, d1b = 0, d2b = 0;
            if (ir.sel == 2) { t1b += dtb; t0b -= dtb; }
            else if (ir.sel == 0) dt0b += 100.0 * dtb;
            else if (!ir.dt1_const) {
                const double mm = ir.max_is_d2 ? (double)ir.d2 : (double)ir.d1;
                const double mb = dtb * (-0.2) * (double)ir.dt1 / mm;
                if (ir.max_is_d2) d2b += mb; else d1b += mb;
            } else if (dt0 * 1e-3f > 1e-6f) dt0b += 1e-3 * dtb;
            const double n2 = (double)ir.d2 * (double)dt0, n2b = d2b / (double)dt0;
            dt0b += -d2b * (double)ir.d2 / (double)dt0;
            const double coef_w = n2 > 0 ? n2b / (N * n2) : 0.0;
            if (writer) { IBState b; b.tb = tb; b.t1b = t1b; b.t0b = t0b; b.dt0b = dt0b; b.d1b = d1b; b.d2b = d2b; b.coef_w = coef_w; b.pad = 0; Bq.ibstate[0] = b; }
            const float cw = (float)coef_w;
            if (tile_ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sk = P.abstol + fabsf(xv[i]) * P.reltol;
                    const float wv = (f1v[i] - f0v[i]) / sk;
                    const float f1b = (colok && r0 + i < P.D ? cw * wv : 0.f) / sk;
                    v[i] = ACT2 ? f1b * (1.f - f1v[i] * f1v[i]) : f1b;
                }
                st4(Bq.zi2 + A + co, r0, P.D, true, vec, v);
            }
        } else {
            // ---- scalars of phase 2 (rnde_binit_kernel, PHASE == 2) ----
            const IBState ib = Bq.ibstate[0];
            double dot1 = 0, tau1 = 0;
            for (int i = lane; i < P.nwg; i += 64) { dot1 += (double)Bq.ipart[4 * i]; tau1 += (double)Bq.ipart[4 * i + 1]; }
            dot1 = wave_sum_d(dot1); tau1 = wave_sum_d(tau1);
            double dt0b = ib.dt0b + tau1 + dot1, t0b = ib.t0b + tau1, t1b = ib.t1b, d1b = ib.d1b, d0b = 0;
            if (ir.dt0_clamped) { t1b += dt0b; t0b -= dt0b; }
            else if (!ir.dt0_const) { d0b = dt0b / (100.0 * (double)ir.d1); d1b += -dt0b * (double)dt0 / (double)ir.d1; }
            const float cv = ir.d1 > 0.f ? (float)(d1b / (N * (double)ir.d1)) : 0.f;
            const float cz = ir.d0 > 0.f ? (float)(d0b / (N * (double)ir.d0)) : 0.f;
            const float cw = (float)ib.coef_w;
            if (writer) { IBState b = ib; b.t1b = t1b; b.t0b = t0b; Bq.ibstate[1] = b; }
            if (tile_ok) {
                f32x4 u0b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = colok && r0 + i < P.D;
                    const float sk = P.abstol + fabsf(xv[i]) * P.reltol;
                    const float wv = (f1v[i] - f0v[i]) / sk, vv = f0v[i] / sk, z = xv[i] / sk;
                    const float wb = ok ? cw * wv : 0.f, vb = ok ? cv * vv : 0.f, zb = ok ? cz * z : 0.f;
                    const float skb = -(wb * wv + vb * vv + zb * z) / sk;
                    const float f0b = K1v[i] + dt0 * ub1[i] + (vb - wb) / sk;
                    u0b[i] = Uv[i] + ub1[i] + zb / sk + skb * P.reltol * sgnf(xv[i]);
                    if (Bq.sv_ubar0 && ok) u0b[i] += Bq.sv_ubar0[((size_t)gcol * Bq.sv_T) * P.D + r0 + i];
                    v[i] = ACT2 ? f0b * (1.f - f0v[i] * f0v[i]) : f0b;
                }
                st4(Bq.zi2 + co, r0, P.D, true, vec, v);
                st_tile(Bq.xbar + co, r0, P.D, colok, false, u0b);      // the direct part of x-bar; STEP 3 adds the part through f0
            }
        }
        // ---- phase D: hbar partial (and layer-2 time row) of this row block -> slab[par] ----
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[col * KG + kperm(16 * w + 4 * (lane >> 4) + i)] = tile_ok ? v[i] : 0.f;
        __syncthreads();
        f32x4* sl = (f32x4*)Q.slab + ((((size_t)par * Q.C + ct) * Q.R + rb) * Q.HT) * 64;
        const float* gbp = GL + col * KG + 4 * (lane >> 4);
        f32x4 bg[kSMaxW];
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb) if (kb < Q.WT) bg[kb] = *(const f32x4*)(gbp + 16 * kb);
        if (w < Q.HT) {
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
        for (int ht = w + Q.WT; ht < Q.HT; ht += Q.WT) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < Q.WT && rb * Q.WT + kb < Q.MT) {
                    const f32x4 a = Q.pwDt[((size_t)ht * Q.MT + rb * Q.WT + kb) * 64 + lane];
                    acc0 = mfma16(a[0], bg[kb][0], acc0);
                    acc1 = mfma16(a[1], bg[kb][1], acc1);
                    acc0 = mfma16(a[2], bg[kb][2], acc0);
                    acc1 = mfma16(a[3], bg[kb][3], acc1);
                }
            }
            sl[(size_t)ht * 64 + lane] = acc0 + acc1;
        }
        return;
    } else {
        // ---- phases A + B of the evaluation (f1 at (u1, t0 + dt0): hidden activations h1; f0 at (u0, t0): h0) ----
        f32x4 wB[kSMaxHT];
#pragma unroll
        for (int kb = 0; kb < kSMaxHT; ++kb)
            if (kb < Q.KHb && tile_ok) wB[kb] = Q.pwBt[((size_t)T * Q.KHb + kb) * 64 + lane];
        const float* hsrc = kPhase == 1 ? P.h1 : P.h0;
        float* z1dst = kPhase == 1 ? Bq.zi1 + HB : Bq.zi1;
        const float* W1t = Q.p + (size_t)P.H * P.D;
        const f32x4* sl = (const f32x4*)Q.slab + (((size_t)par * Q.C + ct) * Q.R) * Q.HT * 64;
        f32x4 f0v = {0.f, 0.f, 0.f, 0.f};
        if (kPhase == 1 && tile_ok) f0v = ld4(P.f0 + co, r0, P.D, true, vec);
        for (int ht = w; ht < Q.HT; ht += Q.WT) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < Q.R; ++r) z += sl[((size_t)r * Q.HT + ht) * 64 + lane];
            const int h0 = 16 * ht + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                float zv = 0.f;
                if (hr < P.H) {
                    const float hv = hsrc[(size_t)gcol * P.H + hr];
                    zv = z[i] * (1.f - hv * hv);
                    if (rb == 0) { z1dst[(size_t)gcol * P.H + hr] = zv; tau += W1t[hr] * zv; }
                } else if (hr == P.H) {
                    if (rb == 0) tau += z[i];          // layer-2 time cotangent (row H of [W2x^T; w2t^T] z2bar)
                }
                if (hr < 16 * Q.KHb) ZL[col * KZ + kperm(hr)] = zv;
            }
        }
        if (Q.KHb > Q.HT) {
            for (int i = tid; i < kSCB * 16 * Q.KHb; i += blockDim.x) {
                const int c = i / (16 * Q.KHb), k = i - c * 16 * Q.KHb;
                if (k >= 16 * Q.HT) ZL[c * KZ + kperm(k)] = 0.f;
            }
        }
        __syncthreads();
        f32x4 gb = {0.f, 0.f, 0.f, 0.f};
        if (tile_ok) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* zb = ZL + col * KZ + 4 * (lane >> 4);
            f32x4 bf[kSMaxHT];
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) if (kb < Q.KHb) bf[kb] = *(const f32x4*)(zb + 16 * kb);
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) {
                if (kb < Q.KHb) {
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
            }
            gb = acc0 + acc1;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (r0 + i >= P.D) gb[i] = 0.f;
            if constexpr (kPhase == 1) {
                st4(Bq.UB1 + co, r0, P.D, true, vec, gb);
#pragma unroll
                for (int i = 0; i < 4; ++i) dot += gb[i] * f0v[i];
            } else {
                const f32x4 u0b = ld_tile(Bq.xbar + co, r0, P.D, colok, false);    // (written by STEP 2 of this same lane's counterpart: same rows, same column)
                st_tile(Bq.xbar + co, r0, P.D, colok, false, u0b + gb);
            }
        }
        if (!colok) tau = 0.f;
        dot = wave_sum_f(dot); tau = wave_sum_f(tau);
        __syncthreads();
        if (lane == 0) { RED[w] = dot; RED[8 + w] = tau; }
        __syncthreads();
        if (tid == 0) {
            float sa = 0.f, ta = 0.f;
            for (int i = 0; i < Q.WT; ++i) { sa += RED[i]; ta += RED[8 + i]; }
            if constexpr (kPhase == 1) { float* o = Bq.ipart + (size_t)wg * 4; o[0] = sa; o[1] = ta; o[2] = 0.f; o[3] = 0.f; }
            else { float* o = Bq.ipart + ((size_t)P.nwg + wg) * 4; o[0] = ta; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; }
        }
    }
}

}  // namespace rnde
