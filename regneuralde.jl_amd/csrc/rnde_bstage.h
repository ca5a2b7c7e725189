// rnde_bstage.h -- stage engine, reverse pass: one launch per reversed Runge-Kutta stage (mirror of rnde_stage.h).
//
// Per attempted step, in reverse (reference: what Tracker.gradient does over the taped solve,
// experiments/mnist_node.jl:229-232 with neural_ode.jl:134):
//   BM_START   scalar adjoint chain; reverse of the error estimate (utilde-bar, seeds of unew-bar / uprev-bar);
//              k7-bar -> z2bar_7; phase D: hbar partial = W2x^T[:, rows] * z2bar[rows]          -> slab
//   BM_STAGE j (j = 6..1, zero-based stage whose f is reversed)
//              phase A: hbar = sum of slabs; z1bar = hbar * (1 - h_j^2); time cotangents
//              phase B: gbar_j[rows] = W1x^T[rows, :] * z1bar
//              phase C: store gbar_j; cotangent of k_{j-1} gathered from the stored gbar_s, unew-bar, utilde-bar;
//                       z2bar_{j-1} (or, for j == 1, the attempt's outputs: uprev-bar and k1-bar)
//              phase D: hbar partial of stage j-1                                                   -> slab
// The cotangent of k_j is gathered (read gbar_s for s > j) instead of scattered, so every array is written once.
#pragma once
#include "rnde_bwd.h"
#include "rnde_stage.h"

namespace rnde {

struct BStageParams {
    BwdParams B;              // shared: F (forward geometry + tape), U, K1, svb_att, bstate, bpart, ubar, n_att, flags
    const float* p;
    const f32x4* pwBt;        // [MT][KHb][64]  W1x^T rows      (phase B)
    const f32x4* pwDt;        // [HT][MT][64]   [W2x^T; w2t^T]  (phase D, K = state rows)
    float* slab;              // [2][C][R][HT][64][4]
    float* UTB; float* UNB; float* UPB0; float* GB;   // utilde-bar, unew-bar, uprev-bar seed, gbar_s (s = 1..6 -> GB + (s-1)*A)
    float* EXK; float* EXG;   // stiffness-estimate extras: direct cotangent of k6, of g6 (regularize >= 2)
    const float* sv_t; const float* sv_ubar; int nsave; float* SVW;   // saveat: times, D x T x B cotangent, W_i = sum_p b_i(theta_p) ubar_p (7 arrays)
    int MT, WT, R, C, HT, KHb;
    const void* x3Bt; const void* x3Dt;   // rnde_x3.h: the two images above split into three bf16 planes (X3 form of rnde_bstage_attempt_kernel), or null
};

enum { BM_START = 0, BM_STAGE = 1 };

template <int ACT2, int MODE>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_bstage_kernel(const BStageParams Q, const int n, const int j, const StepMeta m,
                                                                   const float eig_c1, const float eig_c2, const int sv_lo, const int sv_hi, const double qo_host) {
#pragma clang fp contract(off)   // rounds exactly like rnde_bstage_attempt_kernel (rnde_bstage_persist.h): outputs are compared bit for bit
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
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (P.D & 3) == 0;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    const int T = rb * Q.WT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const bool tile_ok = T < Q.MT;
    const long long A = (long long)P.D * P.Bpad;
    const RecLayout L{A, (long long)P.H * P.Bpad};
    const bool first = (n == Bq.n_att - 1);
    const size_t co = (size_t)gcol * P.D;

    // The attempt's record (which tape slot, dt, flags) is known to the host after the forward pass, so it arrives as
    // a kernel argument: every address below is known at launch and all loads of the launch are issued up front.
    f32x4 wB[kSMaxHT], wD[kSMaxW];
    if constexpr (MODE == BM_STAGE) {
#pragma unroll
        for (int kb = 0; kb < kSMaxHT; ++kb)
            if (kb < Q.KHb && tile_ok) wB[kb] = Q.pwBt[((size_t)T * Q.KHb + kb) * 64 + lane];
    }
    const bool hasD = (MODE == BM_START) || (j > 1);
    if (hasD) {
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb)
            if (kb < Q.WT && w < Q.HT && rb * Q.WT + kb < Q.MT) wD[kb] = Q.pwDt[((size_t)w * Q.MT + rb * Q.WT + kb) * 64 + lane];
    }
    // hbar slabs of the previous launch + the hidden activations they are combined with
    f32x4 zs = {0.f, 0.f, 0.f, 0.f};
    f32x4 zr[kSMaxW];
    float* R = P.arena + (long long)m.rec * P.rec_stride;
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, h_own[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == BM_STAGE) {
        const int par0 = j & 1;
        const f32x4* sl0 = (const f32x4*)Q.slab + (((size_t)par0 * Q.C + ct) * Q.R) * Q.HT * 64;
        if (w < Q.HT) {
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zr[r] = sl0[((size_t)r * Q.HT + w) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hr = 16 * w + 4 * (lane >> 4) + i;
            if (hr < P.H) { w1t_own[i] = Q.p[(size_t)P.H * P.D + hr]; h_own[i] = (R + L.h(j + 1))[(size_t)gcol * P.H + hr]; }
        }
    }

    // phase-C operands of BM_STAGE (unew-bar, utilde-bar, stored gbar_s, k_{j-1}): issue now
    f32x4 c_unb = {0.f, 0.f, 0.f, 0.f}, c_utb = {0.f, 0.f, 0.f, 0.f}, c_ks = {0.f, 0.f, 0.f, 0.f}, c_gs[5];
    if constexpr (MODE == BM_STAGE) {
        if (tile_ok) {
            c_unb = ld4(Q.UNB + co, r0, P.D, true, vec);
            c_utb = ld4(Q.UTB + co, r0, P.D, true, vec);
            if (j >= 2) c_ks = ld4(R + L.k(j) + co, r0, P.D, true, vec);          // k_{j-1} is stored as k(j)
#pragma unroll
            for (int s = 1; s <= 5; ++s) if (s > j || j == 1) c_gs[s - 1] = ld4(Q.GB + (size_t)(s - 1) * A + co, r0, P.D, true, vec);
        }
    }
    const bool accepted = (m.flags & F_ACCEPT) != 0;
    const float dt = m.dt;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (m.src >= 0) { const float* Rl = P.arena + (long long)m.src * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }

    float S = 0.f, tau = 0.f, exdt = 0.f;   // dot partial sum_j <k_j, kbar_j>; time-cotangent partial; extra dt-bar (saveat theta terms)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};

    if constexpr (MODE == BM_START) {
        // ---- scalar adjoint chain, identical in every wave (same code as the column-owner kernel) ----
        double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0;
        if (!first) finish_attempt_scalars(Bq, n + 1, lane, tb, dtpb, qoldb, t1b, t0b);
        float coef;
        {
            const double N = (double)P.D * (double)P.Bn;
            double eb = 0, dtb_pre = 0, q11b = 0, qb = 0, qoldb_in = 0;
            if (accepted) {
                const bool err_term = Bq.reg_kind == 1 || (Bq.reg_kind == 3 && !(m.eest * dt == 0.f));
                if (err_term) { const double sb = (double)Bq.svb_att[n]; eb += sb * (double)dt; dtb_pre += sb * (double)m.eest; }
                if (Bq.reg_kind == 4 && !(m.eigen == 0.f || m.eigen != m.eigen)) dtb_pre += (double)Bq.svb_att[n] * ((double)m.eigen * (double)dt > 0 ? 1.0 : -1.0) * (double)m.eigen;      // |eigen_est * dt|: its dt share (the eigen_est share travels as eig_c1 / eig_c2)
                dtb_pre += tb;
                if (m.flags & F_DTMAXCLAMP) { t1b += dtpb; t0b -= dtpb; }
                else if (Bq.track_ctrl) { dtb_pre += dtpb / (double)m.q; qb += -dtpb * (double)dt / ((double)m.q * (double)m.q); }
                if (m.eest > kQoldInit) eb += qoldb;
            } else {
                dtb_pre += dtpb / (double)m.rej_m;
                if (m.flags & F_REJQ11) q11b += -dtpb * (double)dt / ((double)m.rej_m * (double)m.rej_m) / (double)kGamma;
                qoldb_in = qoldb;
            }
            if (!(m.flags & F_QCLAMP) && !(m.flags & F_EZERO)) {
                const double qo = qo_host;   // = pow(qold_in, beta2), evaluated once on the host (a double pow per wave cost ~1 us of every launch)
                q11b += qb / (qo * (double)kGamma);
                qoldb_in += -(double)kBeta2 * qb * (double)m.q / (double)m.qold_in;
            }
            if (!(m.flags & F_EZERO) && m.eest > 0.f) eb += q11b * (double)kBeta1 * (double)m.q11 / (double)m.eest;
            coef = m.eest > 0.f ? (float)(eb / (N * (double)m.eest)) : 0.f;
            if (writer) { BState b; b.tb_pre = tb; b.dtb_pre = dtb_pre; b.qoldb = qoldb_in; b.t1b = t1b; b.t0b = t0b; b.pad[0] = b.pad[1] = b.pad[2] = 0; Bq.bstate[n & 1] = b; }
        }
        if (tile_ok) {
            const f32x4 upv = ld4(upsrc + co, r0, P.D, upok, upvec);
            const f32x4 unv = ld4(R + L.unew() + co, r0, P.D, true, vec);
            f32x4 kq[7];
            kq[0] = ld4(k1p + co, r0, P.D, true, vec);
#pragma unroll
            for (int s = 2; s <= 7; ++s) kq[s - 1] = ld4(R + L.k(s) + co, r0, P.D, true, vec);
            f32x4 acc = tsBt(0) * kq[0], g6 = tsA(5, 0) * kq[0];
#pragma unroll
            for (int s = 1; s < 7; ++s) { acc += tsBt(s) * kq[s]; if (s < 5) g6 += tsA(5, s) * kq[s]; }
            const f32x4 k6 = kq[5], k7 = kq[6];
            f32x4 uin = {0.f, 0.f, 0.f, 0.f}, k1in = {0.f, 0.f, 0.f, 0.f};
            const bool sv_mode = Q.nsave > 0;          // saveat: the only outputs are the saved points, no u_end cotangent
            if (accepted) {
                if (!first) { uin = ld4(Bq.U + co, r0, P.D, true, vec); k1in = ld4(Bq.K1 + co, r0, P.D, true, vec); }
                else if (!sv_mode) uin = ld4(Bq.ubar + co, r0, P.D, colok, false);
            }
            f32x4 utb, unb, upb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ut = dt * acc[i];
                const float au = fabsf(upv[i]), an = fabsf(unv[i]);
                const bool use_new = !(au > an);
                const float sk = P.abstol + (use_new ? an : au) * P.reltol;
                const float r = ut / sk;
                const float rb_ = colok ? coef * r : 0.f;
                const float skb = -rb_ * r / sk;
                utb[i] = rb_ / sk;
                unb[i] = uin[i] + (use_new ? skb * P.reltol * sgnf(unv[i]) : 0.f);
                upb[i] = use_new ? 0.f : skb * P.reltol * sgnf(upv[i]);
            }
            f32x4 w7 = {0.f, 0.f, 0.f, 0.f};
            if (sv_hi > sv_lo) {
                // reverse of the dense output u(ts) = uprev + dt sum_i b_i(theta) k_i, theta = (ts - t)/dt  
                f32x4 Wv[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) Wv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float tnew = m.t + dt;
                for (int idx = sv_lo; idx < sv_hi; ++idx) {
                    const float ts = Q.sv_t[idx];
                    const f32x4 ub = ld4(Q.sv_ubar + ((size_t)gcol * Q.nsave + idx) * P.D, r0, P.D, colok, vec);
                    if (ts == tnew) { unb += ub; continue; }
                    const float th = (ts - m.t) / dt;
                    float bw[7], dbw[7];
                    dense_weights(th, bw);
                    dense_weights_deriv(th, dbw);
                    upb += ub;
                    f32x4 dacc = dbw[0] * kq[0];
#pragma unroll
                    for (int i = 0; i < 7; ++i) { Wv[i] += bw[i] * ub; if (i) dacc += dbw[i] * kq[i]; }
                    float dth = 0.f;                 // <ubar_p, dt * sum_i b_i'(theta) k_i>
#pragma unroll
                    for (int i = 0; i < 4; ++i) dth += ub[i] * dt * dacc[i];
                    tau += -dth / dt;                // t-bar    (theta = (ts - t)/dt)
                    exdt += -dth * th / dt;          // dt-bar, beyond the dt-scaled part that S carries
                }
#pragma unroll
                for (int i = 0; i < 7; ++i) st4(Q.SVW + (size_t)i * A + co, r0, P.D, true, vec, Wv[i]);
                w7 = Wv[6];
            }
            st4(Q.UTB + co, r0, P.D, true, vec, utb);
            st4(Q.UNB + co, r0, P.D, true, vec, unb);
            st4(Q.UPB0 + co, r0, P.D, true, vec, upb);
            f32x4 kb7 = dt * (tsBt(6) * utb + w7);
#pragma unroll
            for (int i = 0; i < 4; ++i) S += k7[i] * kb7[i];
            kb7 += k1in;
            if (eig_c1 != 0.f || eig_c2 != 0.f) {
                // reverse of eigen_est = ||k7-k6|| / ||unew-g6|| (direct terms; not part of S: they do not scale with dt)
                f32x4 exk, exg;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = colok && (r0 + i < P.D);
                    const float d1 = k7[i] - k6[i], d2 = unv[i] - (upv[i] + dt * g6[i]);
                    kb7[i] += ok ? eig_c1 * d1 : 0.f;
                    exk[i] = ok ? -eig_c1 * d1 : 0.f;
                    unb[i] += ok ? eig_c2 * d2 : 0.f;
                    exg[i] = ok ? -eig_c2 * d2 : 0.f;
                }
                st4(Q.UNB + co, r0, P.D, true, vec, unb);
                st4(Q.EXK + co, r0, P.D, true, vec, exk);
                st4(Q.EXG + co, r0, P.D, true, vec, exg);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (r0 + i < P.D) ? (ACT2 ? kb7[i] * (1.f - k7[i] * k7[i]) : kb7[i]) : 0.f;
            st4(R + L.k(7) + co, r0, P.D, true, vec, v);   // z2bar_7 replaces k7 (dead from here on)
        }
    } else {
        // ---- phase A: hbar from the slabs; z1bar; time cotangents ----
        if (w < Q.HT) {
            const f32x4* sl0 = (const f32x4*)Q.slab + (((size_t)(j & 1) * Q.C + ct) * Q.R) * Q.HT * 64;
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zs += zr[r];
            for (int r = kSMaxW; r < Q.R; ++r) zs += sl0[((size_t)r * Q.HT + w) * 64 + lane];
        }
        const float* W1t = Q.p + (size_t)P.H * P.D;
        const f32x4* sl = (const f32x4*)Q.slab + (((size_t)(j & 1) * Q.C + ct) * Q.R) * Q.HT * 64;
        const float* hsrc = R + L.h(j + 1);
        float* z1dst = R + L.z1(j + 1);
        for (int ht = w; ht < Q.HT; ht += Q.WT) {
            f32x4 z = zs;
            if (ht != w) {
                z = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < Q.R; ++r) z += sl[((size_t)r * Q.HT + ht) * 64 + lane];
            }
            const int h0 = 16 * ht + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                float zv = 0.f;
                if (hr < P.H) {
                    const float hv = (ht == w) ? h_own[i] : hsrc[(size_t)gcol * P.H + hr];
                    zv = z[i] * (1.f - hv * hv);
                    if (rb == 0) { z1dst[(size_t)gcol * P.H + hr] = zv; tau += ((ht == w) ? w1t_own[i] : W1t[hr]) * zv; }
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
        // ---- phase B: gbar_j for this wave's 16 rows ----
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
        }
        // ---- phase C ----
        if (tile_ok) {
            const bool has_eig = (eig_c1 != 0.f || eig_c2 != 0.f);
            if (has_eig && j == 5) gb += ld4(Q.EXG + co, r0, P.D, true, vec);        // direct cotangent of g6 joins gbar of stage 6
            st4(Q.GB + (size_t)(j - 1) * A + co, r0, P.D, true, vec, gb);
            f32x4 unb = c_unb;
            if (j == 6) { unb += gb; st4(Q.UNB + co, r0, P.D, true, vec, unb); }
            const f32x4 utb = c_utb;
            const int jn = j - 1;                                   // zero-based index of the k whose cotangent is now complete
            f32x4 kbar = tsA_rt(6, jn) * unb + kTsBt[jn] * utb;
#pragma unroll
            for (int s = 1; s <= 5; ++s) {
                if (s > jn) kbar += tsA_rt(s, jn) * ((s == j) ? gb : c_gs[s - 1]);
            }
            if (sv_hi > sv_lo) kbar += ld4(Q.SVW + (size_t)jn * A + co, r0, P.D, true, vec);   // saveat: dense-output weights on k_jn
            kbar = dt * kbar;
            if (jn >= 1) {
                const f32x4 ks = c_ks;
#pragma unroll
                for (int i = 0; i < 4; ++i) S += ks[i] * kbar[i];
                if (has_eig && j == 6) kbar += ld4(Q.EXK + co, r0, P.D, true, vec);   // direct cotangent of k6 (after S: not dt-scaled)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (r0 + i < P.D) ? (ACT2 ? kbar[i] * (1.f - ks[i] * ks[i]) : kbar[i]) : 0.f;
                st4(R + L.k(jn + 1) + co, r0, P.D, true, vec, v);  // z2bar replaces k (dead)
            } else {
                // j == 1: kbar is the cotangent of k1; assemble the attempt's outputs
                const f32x4 k1v = ld4(k1p + co, r0, P.D, true, vec);
#pragma unroll
                for (int i = 0; i < 4; ++i) S += k1v[i] * kbar[i];
                f32x4 uo = ld4(Q.UPB0 + co, r0, P.D, true, vec) + unb;
#pragma unroll
                for (int s = 1; s <= 5; ++s) uo += (s == j) ? gb : c_gs[s - 1];
                f32x4 ko = kbar;
                if (!accepted) {
                    uo += first ? ld4(Bq.ubar + co, r0, P.D, colok, false) : ld4(Bq.U + co, r0, P.D, true, vec);
                    if (!first) ko += ld4(Bq.K1 + co, r0, P.D, true, vec);
                }
                st4(Bq.U + co, r0, P.D, true, vec, uo);
                st4(Bq.K1 + co, r0, P.D, true, vec, ko);
            }
        }
    }

    if (hasD) {
        // ---- phase D: hbar partial (and layer-2 time row) of this row block ----
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[col * KG + kperm(16 * w + 4 * (lane >> 4) + i)] = tile_ok ? v[i] : 0.f;
        __syncthreads();
        const int par = (MODE == BM_START) ? 0 : ((j - 1) & 1);   // consumed by BM_STAGE(j') with par0 = j' & 1 (START feeds j' = 6)
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
    }

    // ---- per-workgroup partials {S, tau, c_j * tau} accumulated over the 7 launches of the attempt ----
    if (!colok) { tau = 0.f; exdt = 0.f; }
    S = wave_sum_f(S); tau = wave_sum_f(tau); exdt = wave_sum_f(exdt);
    __syncthreads();
    if (lane == 0) { RED[w] = S; RED[8 + w] = tau; RED[16 + w] = exdt; }
    __syncthreads();
    if (tid == 0) {
        float sa = 0.f, ta = 0.f, xa = 0.f;
        for (int i = 0; i < Q.WT; ++i) { sa += RED[i]; ta += RED[8 + i]; xa += RED[16 + i]; }
        float* o = Bq.bpart + ((size_t)(n & 1) * Bq.bpart_n + blockIdx.x) * 4;
        if constexpr (MODE == BM_START) { o[0] = sa; o[1] = ta; o[2] = xa; o[3] = 0.f; }   // (saveat theta terms: t-bar, dt-bar)
        else { o[0] += sa; o[1] += ta; o[2] += kTsC[j] * ta; }
    }
}

}  // namespace rnde
