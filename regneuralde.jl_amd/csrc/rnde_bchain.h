// rnde_bchain.h -- reverse pass of the chain engine (rnde_chain.h): discretise-then-optimise through the taped Tsit5
// attempts of a small-width Dense chain, same scalar chain (t, dt, EEst, PI controller, initial-step heuristic) as
// rnde_bwd.h, whose BState/IBState/finish_attempt_scalars are reused.
//
//   rnde_bchain_kernel       one attempted step, reverse; one wave owns the same 16 batch columns as in the forward,
//                            cotangents of k1..k7 / uprev stay in registers.  J_f^T products: the layer inputs are
//                            recomputed forward (MFMA chain), then the transposed fragments run the chain backwards.
//   rnde_bchain_init_kernel  reverse of k1 = f(u0,t0) and of the initial-step heuristic (2 phases).
//   rnde_chain_wgrad_kernel  parameter gradients of ALL layers over ALL f evaluations of the solve, off the sweep:
//                            the sweep only dumps (layer input, pre-activation cotangent) pairs in fragment order;
//                            this kernel contracts them over batch columns x evaluations with 16x16x4 MFMAs
//                            (deterministic: per-chunk partials, fixed-order reduction by rnde_wgrad_reduce).
#pragma once
#include "rnde_chain.h"
#include "rnde_bwd.h"

namespace rnde {

struct BChainParams {
    BwdParams B;            // B.U / B.K1 / B.UB1 are fragment-order arrays here
    ChainGeo G;
    const float* frags;
    int ntiles;
    float* slab;            // [n_evals][ntiles][RS][64]: per layer its input (H) and its pre-activation cotangent (Z)
    long long ev_stride;    // ntiles * RS * 64
    int RS;                 // k-step rows per (evaluation, tile)
    int hrow[kCMaxL + 1], zrow[kCMaxL];   // rows padded to whole tiles; hrow[n_layers] = the evaluation's output
    const float* sv_t; const float* sv_ubar; int nsave;   // saveat: times, D x T x B cotangent (caller layout)
};

template <int MT>
__device__ __forceinline__ void chain_store_rows_t(float* p, const float (&a)[kCMaxKs]) {
#pragma unroll
    for (int ks = 0; ks < 4 * MT; ++ks) p[ks * 64] = a[ks];
}
__device__ __forceinline__ void chain_store_rows(float* p, int mt, const float (&a)[kCMaxKs]) {
    switch (mt) {
        case 1: chain_store_rows_t<1>(p, a); break;  case 2: chain_store_rows_t<2>(p, a); break;
        case 3: chain_store_rows_t<3>(p, a); break;  default: chain_store_rows_t<4>(p, a); break;
    }
}
// z = abar .* act'(out) for the MT output tiles of a layer; dumps z, accumulates the time-column cotangent
template <int MT>
__device__ __forceinline__ void chain_zstage_t(const float (&ab)[kCMaxKs], bool th, const float* op, float* zp, const float* wt, int td,
                                               float (&z)[kCMaxKs], float& tl) {
    float o[4 * MT];
    if (th) {
#pragma unroll
        for (int ks = 0; ks < 4 * MT; ++ks) o[ks] = op[ks * 64];
    }
#pragma unroll
    for (int ks = 0; ks < kCMaxKs; ++ks) {
        float v = 0.f;
        if (ks < 4 * MT) { v = ab[ks]; if (th) v *= (1.f - o[ks < 4 * MT ? ks : 0] * o[ks < 4 * MT ? ks : 0]); zp[ks * 64] = v; }
        z[ks] = v;
    }
    if (td) {
        float w[4 * MT];
#pragma unroll
        for (int ks = 0; ks < 4 * MT; ++ks) w[ks] = wt[ks * 64];
#pragma unroll
        for (int ks = 0; ks < 4 * MT; ++ks) tl = fmaf(z[ks], w[ks], tl);
    }
}

// J_f^T product for the wave's 16 columns at the point (g, ts) whose value kout = f(g, ts) is on the tape.
// Dumps every layer's input and pre-activation cotangent for the weight-gradient kernel; returns gbar and adds the
// cotangent of the time input (TDChain layers) to tau.
template <int NKD, int ALT = 0>
__device__ __forceinline__ void chain_fbwd(const BChainParams& Q, const float* FR, const float* BF, const float* TF, float ts,
                                           const float (&g)[NKD], const float (&kout)[NKD], const float (&kbar)[NKD], float (&gb)[NKD],
                                           float* __restrict__ sl, float& tau, int lane) {
    const ChainGeo& G = Q.G;
    float a[kCMaxKs];
#pragma unroll
    for (int k = 0; k < kCMaxKs; ++k) a[k] = (k < NKD) ? (G.pre_act ? tanh_fast(g[k < NKD ? k : 0]) : g[k < NKD ? k : 0]) : 0.f;
    // forward recompute: every layer's input goes to the slab; the last layer's output is kout (taped)
#pragma unroll 1
    for (int l = 0; l < G.n_layers; ++l) {
        if constexpr (ALT == 1) {
            if ((l & 1) == 0) chain_store_rows_t<(kAltA + 3) / 4>(sl + (size_t)Q.hrow[l] * 64, a);
            else chain_store_rows_t<(kAltB + 3) / 4>(sl + (size_t)Q.hrow[l] * 64, a);
        } else chain_store_rows(sl + (size_t)Q.hrow[l] * 64, (G.nks[l] + 3) >> 2, a);
        if (l + 1 < G.n_layers) chain_layer<ALT>(G, FR, BF, l, ts, a, lane);
    }
    float ab[kCMaxKs];
#pragma unroll
    for (int k = 0; k < kCMaxKs; ++k) { a[k] = (k < NKD) ? kout[k < NKD ? k : 0] : 0.f; ab[k] = (k < NKD) ? kbar[k < NKD ? k : 0] : 0.f; }
    chain_store_rows(sl + (size_t)Q.hrow[G.n_layers] * 64, (G.nks[G.n_layers] + 3) >> 2, a);
    float tl = 0.f;
#pragma unroll 1
    for (int l = G.n_layers - 1; l >= 0; --l) {
        const int nout = G.nks[l + 1], mto = (nout + 3) >> 2, mti = (G.nks[l] + 3) >> 2;
        const bool th = G.act[l] != 0;
        const float* op = sl + (size_t)Q.hrow[l + 1] * 64;   // layer output = next layer's input (or kout)
        float* zp = sl + (size_t)Q.zrow[l] * 64;
        const float* wt = BF + (size_t)(G.boff[l] + 4 * mto) * 64 + lane;
        float z[kCMaxKs];
        if constexpr (ALT == 1) {
            constexpr int MA = (kAltA + 3) / 4, MB = (kAltB + 3) / 4;
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((l & 1) == 0) {   // forward kAltA -> kAltB: z has kAltB k-steps, the product kAltA
                chain_zstage_t<MB>(ab, th, op, zp, wt, G.time_dep, z, tl);
                chain_mm_t<kAltB>(TF + (size_t)G.toff[l] * 64 + lane, MA, z, acc);
                chain_act_t<MA>(acc, false, ab);
            } else {
                chain_zstage_t<MA>(ab, th, op, zp, wt, G.time_dep, z, tl);
                chain_mm_t<kAltA>(TF + (size_t)G.toff[l] * 64 + lane, MB, z, acc);
                chain_act_t<MB>(acc, false, ab);
            }
            continue;
        }
        switch (mto) {
            case 1: chain_zstage_t<1>(ab, th, op, zp, wt, G.time_dep, z, tl); break;
            case 2: chain_zstage_t<2>(ab, th, op, zp, wt, G.time_dep, z, tl); break;
            case 3: chain_zstage_t<3>(ab, th, op, zp, wt, G.time_dep, z, tl); break;
            default: chain_zstage_t<4>(ab, th, op, zp, wt, G.time_dep, z, tl); break;
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        chain_mm(TF + (size_t)G.toff[l] * 64 + lane, nout, mti, z, acc);
        switch (mti) {
            case 1: chain_act_t<1>(acc, false, ab); break;  case 2: chain_act_t<2>(acc, false, ab); break;
            case 3: chain_act_t<3>(acc, false, ab); break;  default: chain_act_t<4>(acc, false, ab); break;
        }
    }
#pragma unroll
    for (int k = 0; k < NKD; ++k) {
        float v = ab[k];
        if (G.pre_act) { const float a0 = tanh_fast(g[k]); v *= (1.f - a0 * a0); }
        gb[k] = v;
    }
    tau += tl;
}

template <int NKD, int ALT = 0>
__global__ __launch_bounds__(64 * kCW) void rnde_bchain_kernel(const BChainParams Q, const int n, const StepMeta m, const int sv_lo, const int sv_hi, const float eig_c1, const float eig_c2) {
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    const ChainGeo& G = Q.G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FR = smem;
    float* BF = FR + (size_t)G.nfrag_f * 64;
    float* TF = BF + (size_t)G.nfrag_b * 64;
    const int fill_units = (G.nfrag_f + G.nfrag_b + G.nfrag_t + 3) >> 2;
    float* RED = smem + (size_t)fill_units * 256;   // [3][kCW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x * kCW + wave;
    const bool tile_ok = tile < Q.ntiles;
    const int col = lane & 15, g = lane >> 4, gcol = tile * 16 + col;
    const bool colok = tile_ok && gcol < P.B;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    const bool first = (n == Bq.n_att - 1);
    constexpr int nksD = NKD;   // arena arrays are padded to NKD k-steps
    const ChainRec L{(long long)Q.ntiles * NKD * 64};
    const size_t fo = ((size_t)tile * NKD) * 64 + lane;
    chain_fill_lds(Q.frags, smem, fill_units, wave, lane);
    // ---- scalar chain, identical in every wave; same arithmetic as rnde_bstep_kernel ----
    double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0;
    if (!first) finish_attempt_scalars(Bq, n + 1, lane, tb, dtpb, qoldb, t1b, t0b);
    const bool accepted = (m.flags & F_ACCEPT) != 0;
    const float dt = m.dt, t = m.t;
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
            const double qo = pow((double)m.qold_in, (double)kBeta2);
            q11b += qb / (qo * (double)kGamma);
            qoldb_in += -(double)kBeta2 * qb * (double)m.q / (double)m.qold_in;
        }
        if (!(m.flags & F_EZERO) && m.eest > 0.f) eb += q11b * (double)kBeta1 * (double)m.q11 / (double)m.eest;
        coef = m.eest > 0.f ? (float)(eb / (N * (double)m.eest)) : 0.f;
        if (writer) { BState b; b.tb_pre = tb; b.dtb_pre = dtb_pre; b.qoldb = qoldb_in; b.t1b = t1b; b.t0b = t0b; b.pad[0] = b.pad[1] = b.pad[2] = 0; Bq.bstate[n & 1] = b; }
    }

    float S = 0.f, tau = 0.f, ctau = 0.f;   // sum_j <k_j, kbar_j>; sum of time cotangents; c_s-weighted (+ extra dt-bar)
    if (tile_ok) {
        const float* R = P.arena + (long long)m.rec * P.rec_stride;
        float* sl0 = Q.slab + (size_t)(6 * n) * Q.ev_stride + ((size_t)tile * Q.RS) * 64 + lane;
        const bool sv_mode = Q.nsave > 0;      // saveat: the only outputs are the saved points
        float utb[NKD], unb[NKD], upb[NKD], k1v[NKD], Wv[7][NKD];
        // ---- A: reverse of the error estimate; seeds of unew-bar / uprev-bar ----
        {
            float kq[7][NKD], upv[NKD], unv[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool in = q < nksD;
                upv[q] = in ? R[L.upc() + fo + q * 64] : 0.f;
                unv[q] = in ? R[L.unew() + fo + q * 64] : 0.f;
                kq[0][q] = in ? R[L.k1c() + fo + q * 64] : 0.f;
#pragma unroll
                for (int j = 1; j < 7; ++j) kq[j][q] = in ? R[L.k(j + 1) + fo + q * 64] : 0.f;
                k1v[q] = kq[0][q];
            }
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool valid = colok && q < nksD && 4 * q + g < P.D;
                float acc = tsBt(0) * kq[0][q];
#pragma unroll
                for (int j = 1; j < 7; ++j) acc += tsBt(j) * kq[j][q];
                float uin = 0.f;
                if (accepted && q < nksD) {
                    if (!first) uin = Bq.U[fo + q * 64];
                    else if (!sv_mode) uin = ldc(Bq.ubar, P.D, gcol, 4 * q + g, colok);
                }
                utb[q] = 0.f; unb[q] = uin; upb[q] = 0.f;
                if (valid) {
                    const float ut = dt * acc;
                    const float au = fabsf(upv[q]), an = fabsf(unv[q]);
                    const bool use_new = !(au > an);
                    const float sk = P.abstol + (use_new ? an : au) * P.reltol;
                    const float r = ut / sk;
                    const float rb = coef * r;
                    const float skb = -rb * r / sk;
                    utb[q] = rb / sk;
                    if (use_new) unb[q] += skb * P.reltol * sgnf(unv[q]); else upb[q] = skb * P.reltol * sgnf(upv[q]);
                }
#pragma unroll
                for (int i = 0; i < 7; ++i) Wv[i][q] = 0.f;
            }
            if (sv_hi > sv_lo) {
                // reverse of the dense output u(ts) = uprev + dt sum_i b_i(theta) k_i, theta = (ts - t)/dt  
                const float tnew = m.t + dt;
                for (int idx = sv_lo; idx < sv_hi; ++idx) {
                    const float ts = Q.sv_t[idx];
                    const bool at_end = (ts == tnew);
                    const float th = (ts - m.t) / dt;
                    float bw[7], dbw[7];
                    dense_weights(th, bw);
                    dense_weights_deriv(th, dbw);
                    float dth = 0.f;
#pragma unroll
                    for (int q = 0; q < NKD; ++q) {
                        const float ub = (q < nksD && colok && 4 * q + g < P.D) ? Q.sv_ubar[((size_t)gcol * Q.nsave + idx) * P.D + 4 * q + g] : 0.f;
                        if (at_end) unb[q] += ub;
                        else {
                            upb[q] += ub;
                            float dacc = dbw[0] * kq[0][q];
#pragma unroll
                            for (int i = 0; i < 7; ++i) { Wv[i][q] += bw[i] * ub; if (i) dacc += dbw[i] * kq[i][q]; }
                            dth += ub * dt * dacc;
                        }
                    }
                    if (!at_end) { tau += -dth / dt; ctau += -dth * th / dt; }
                }
            }
        }
        // Rb[i] = cotangent of k_{s-i} (zero-based) where s is the next stage to be reversed (rolled loop, kBwdShift)
        float Rb[6][NKD], gb[NKD], exk[NKD], exg[NKD];
        const bool has_eig = (eig_c1 != 0.f || eig_c2 != 0.f);
        // ---- B: stage 7 (k7 = f(unew, t + dt)) ----
        {
            float k7[NKD], unv[NKD], kb7[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool in = q < nksD;
                k7[q] = in ? R[L.k(7) + fo + q * 64] : 0.f;
                unv[q] = in ? R[L.unew() + fo + q * 64] : 0.f;
                kb7[q] = dt * (tsBt(6) * utb[q] + Wv[6][q]);
                S += k7[q] * kb7[q];
                if (accepted && !first && in) kb7[q] += Bq.K1[fo + q * 64];
                exk[q] = 0.f; exg[q] = 0.f;
                if (has_eig) {   // reverse of eigen_est = ||k7-k6|| / ||unew-g6|| (direct terms: they do not scale with dt, so not in S)
                    const bool ok = colok && 4 * q + g < P.D;
                    const float d1 = ok ? k7[q] - R[L.k(6) + fo + q * 64] : 0.f, d2 = ok ? unv[q] - R[L.g(6) + fo + q * 64] : 0.f;
                    kb7[q] += eig_c1 * d1; exk[q] = -eig_c1 * d1;
                    unb[q] += eig_c2 * d2; exg[q] = -eig_c2 * d2;
                }
            }
            float t7 = 0.f;
            chain_fbwd<NKD, ALT>(Q, FR, BF, TF, t + dt, unv, k7, kb7, gb, sl0 + 5 * Q.ev_stride, t7, lane);
            tau += t7; ctau += t7;
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                unb[q] += gb[q];
#pragma unroll
                for (int i = 0; i < 6; ++i) Rb[i][q] = dt * (tsA(6, 5 - i) * unb[q] + tsBt(5 - i) * utb[q] + Wv[5 - i][q]);   // kbar_{5-i}
                upb[q] += unb[q];
            }
        }
        // ---- C: stages 6..2 ----
#pragma unroll 1
        for (int s = 5; s >= 1; --s) {   // zero-based: k_s = f(g_s, t + c_s dt), taped as k(s+1), g(s+1)
            float ks[NKD], gs[NKD], kb[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool in = q < nksD;
                ks[q] = in ? R[L.k(s + 1) + fo + q * 64] : 0.f;
                gs[q] = in ? R[L.g(s + 1) + fo + q * 64] : 0.f;
                kb[q] = Rb[0][q];
                S += ks[q] * kb[q];
                if (has_eig && s == 5) kb[q] += exk[q];          // direct cotangent of k6
            }
            float ts_ = 0.f;
            chain_fbwd<NKD, ALT>(Q, FR, BF, TF, t + kTsC[s] * dt, gs, ks, kb, gb, sl0 + (size_t)(s - 1) * Q.ev_stride, ts_, lane);
            tau += ts_; ctau += kTsC[s] * ts_;
            if (has_eig && s == 5) {
#pragma unroll
                for (int q = 0; q < NKD; ++q) gb[q] += exg[q];   // direct cotangent of g6
            }
            float cb[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) cb[i] = dt * kBwdShift[s][i];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
#pragma unroll
                for (int i = 0; i < 5; ++i) Rb[i][q] = Rb[i + 1][q] + cb[i] * gb[q];
                upb[q] += gb[q];
            }
        }
        // ---- D: k1 and outputs (Rb[0] is now the cotangent of k1) ----
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            S += k1v[q] * Rb[0][q];
            if (q < nksD) {
                float uo = upb[q], ko = Rb[0][q];
                if (!accepted) {
                    if (!first) { uo += Bq.U[fo + q * 64]; ko += Bq.K1[fo + q * 64]; }
                    else if (!sv_mode) uo += ldc(Bq.ubar, P.D, gcol, 4 * q + g, colok);
                }
                Bq.U[fo + q * 64] = uo;
                Bq.K1[fo + q * 64] = ko;
            }
        }
        if (!colok) { tau = 0.f; ctau = 0.f; }
    }
    S = wave_sum_f(S); tau = wave_sum_f(tau); ctau = wave_sum_f(ctau);
    if (lane == 0) { RED[wave] = S; RED[kCW + wave] = tau; RED[2 * kCW + wave] = ctau; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, ta = 0.f, ca = 0.f;
        for (int w = 0; w < kCW; ++w) { s += RED[w]; ta += RED[kCW + w]; ca += RED[2 * kCW + w]; }
        float* o = Bq.bpart + ((size_t)(n & 1) * Bq.bpart_n + blockIdx.x) * 4;
        o[0] = s; o[1] = ta; o[2] = ca; o[3] = 0.f;
    }
}

// Reverse of the initialisation (mirror of rnde_binit_kernel): PHASE 1 = f1 = f(u1, t0 + dt0) of the initial-step
// heuristic, PHASE 2 = f0 = f(u0, t0) (fsalfirst and the heuristic's first evaluation) and x-bar.
template <int NKD, int PHASE, int ALT = 0>
__global__ __launch_bounds__(64 * kCW) void rnde_bchain_init_kernel(const BChainParams Q) {
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    const ChainGeo& G = Q.G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FR = smem;
    float* BF = FR + (size_t)G.nfrag_f * 64;
    float* TF = BF + (size_t)G.nfrag_b * 64;
    const int fill_units = (G.nfrag_f + G.nfrag_b + G.nfrag_t + 3) >> 2;
    float* RED = smem + (size_t)fill_units * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x * kCW + wave;
    const bool tile_ok = tile < Q.ntiles;
    const int col = lane & 15, g = lane >> 4, gcol = tile * 16 + col;
    const bool colok = tile_ok && gcol < P.B;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    constexpr int nksD = NKD;
    const size_t fo = ((size_t)tile * NKD) * 64 + lane;
    const double N = (double)P.D * (double)P.Bn;
    chain_fill_lds(Q.frags, smem, fill_units, wave, lane);
    const InitRec ir = *P.initrec;
    const float dt0 = ir.dt0;
    float* sl = Q.slab + (size_t)(6 * Bq.n_att + (PHASE == 1 ? 1 : 0)) * Q.ev_stride + ((size_t)tile * Q.RS) * 64 + lane;
    float dot = 0.f, tau = 0.f;
    if constexpr (PHASE == 1) {
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
            float f0v[NKD], f1v[NKD], u1v[NKD], f1b[NKD], gb[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool in = q < nksD, valid = colok && in && 4 * q + g < P.D;
                const float xv = ldc(P.x, P.D, gcol, 4 * q + g, colok && in);
                f0v[q] = in ? P.f0[fo + q * 64] : 0.f; f1v[q] = in ? P.f1[fo + q * 64] : 0.f; u1v[q] = in ? P.u1[fo + q * 64] : 0.f;
                f1b[q] = 0.f;
                if (valid) { const float sk = P.abstol + fabsf(xv) * P.reltol; f1b[q] = cw * ((f1v[q] - f0v[q]) / sk) / sk; }
            }
            chain_fbwd<NKD, ALT>(Q, FR, BF, TF, P.t0 + dt0, u1v, f1v, f1b, gb, sl, tau, lane);
#pragma unroll
            for (int q = 0; q < NKD; ++q) { if (q < nksD) Bq.UB1[fo + q * 64] = gb[q]; dot += gb[q] * f0v[q]; }
            if (!colok) tau = 0.f;
        }
        dot = wave_sum_f(dot); tau = wave_sum_f(tau);
        if (lane == 0) { RED[wave] = dot; RED[kCW + wave] = tau; }
        __syncthreads();
        if (tid == 0) {
            float s = 0.f, ta = 0.f;
            for (int w = 0; w < kCW; ++w) { s += RED[w]; ta += RED[kCW + w]; }
            float* o = Bq.ipart + (size_t)blockIdx.x * 4;
            o[0] = s; o[1] = ta; o[2] = 0.f; o[3] = 0.f;
        }
    } else {
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
            float xq[NKD], f0v[NKD], f0b[NKD], u0b[NKD], gb[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const bool in = q < nksD, valid = colok && in && 4 * q + g < P.D;
                xq[q] = ldc(P.x, P.D, gcol, 4 * q + g, colok && in);
                f0v[q] = in ? P.f0[fo + q * 64] : 0.f;
                const float f1v = in ? P.f1[fo + q * 64] : 0.f, ub1 = in ? Bq.UB1[fo + q * 64] : 0.f;
                const float Uv = in ? Bq.U[fo + q * 64] : 0.f, K1v = in ? Bq.K1[fo + q * 64] : 0.f;
                f0b[q] = K1v + dt0 * ub1; u0b[q] = Uv + ub1;
                if (valid) {
                    const float sk = P.abstol + fabsf(xq[q]) * P.reltol;
                    const float w = (f1v - f0v[q]) / sk, v = f0v[q] / sk, z = xq[q] / sk;
                    const float wb = cw * w, vb = cv * v, zb = cz * z;
                    const float skb = -(wb * w + vb * v + zb * z) / sk;
                    f0b[q] += (vb - wb) / sk;
                    u0b[q] += zb / sk + skb * P.reltol * sgnf(xq[q]);
                    if (Bq.sv_ubar0) u0b[q] += Bq.sv_ubar0[((size_t)gcol * Bq.sv_T) * P.D + 4 * q + g];
                }
            }
            chain_fbwd<NKD, ALT>(Q, FR, BF, TF, P.t0, xq, f0v, f0b, gb, sl, tau, lane);
#pragma unroll
            for (int q = 0; q < NKD; ++q)
                if (q < nksD && colok && 4 * q + g < P.D) Bq.xbar[(size_t)gcol * P.D + 4 * q + g] = u0b[q] + gb[q];
            if (!colok) tau = 0.f;
        }
        tau = wave_sum_f(tau);
        if (lane == 0) RED[wave] = tau;
        __syncthreads();
        if (tid == 0) {
            float ta = 0.f;
            for (int w = 0; w < kCW; ++w) ta += RED[w];
            float* o = Bq.ipart + ((size_t)P.nwg + blockIdx.x) * 4;
            o[0] = ta; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        }
    }
}

// ---- parameter gradients: W_l-bar[o][i] = sum over (evaluation, column) of Z_l[o] * [H_l ; t ; 1][i] ---------------------
// grid (n_layers, chunks); each wave contracts its share of (evaluation, tile) units, 16 columns = 4 MFMA k-steps each.
static __global__ __launch_bounds__(64 * kCW) void rnde_chain_wgrad_kernel(const BChainParams Q, const float* __restrict__ ev_t, int n_units, int per_chunk,
                                                                   float* __restrict__ wslab, int P_total) {
    const ChainGeo& G = Q.G;
    __shared__ __attribute__((aligned(16))) float ACC[20 * 256];
    const int l = blockIdx.x, chunk = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int in = G.width[l], out = G.width[l + 1], nin = G.nks[l], nout = G.nks[l + 1];
    const int it = (nin + 3) >> 2, ot = (nout + 3) >> 2;
    const int rho = lane & 15, kk = lane >> 4;
    f32x4 acc[4][5];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int u_lo = chunk * per_chunk, u_hi = min(n_units, u_lo + per_chunk);
    // A unit's 16 columns are contracted in four k-steps; WHICH four columns share a k-step is free as long as both operands agree, so
    // lane (rho, kk) takes the float4 of columns 4 kk .. 4 kk + 3 of its row -- one 16-byte load per tile, a tile's 16 rows 1 KiB of
    // consecutive memory -- where it used to gather column 4 s + kk per k-step with sixteen 4-byte loads; and the next unit's eight loads
    // are in flight while this unit's up to 80 MFMAs issue (the loop used to wait for cold memory four times per unit: the kernel ran at
    // a third of what reading the (H, Z) dump once costs).
    typedef const __attribute__((address_space(1))) f32x4* gq4;
    auto load_unit = [&](int u, f32x4 (&za)[4], f32x4 (&hb)[4], float& te) {
        const int e = u / Q.ntiles, tile = u - e * Q.ntiles;
        const float* base = Q.slab + (size_t)e * Q.ev_stride + ((size_t)tile * Q.RS) * 64;
        const float* Zp = base + (size_t)Q.zrow[l] * 64 + rho * 16 + 4 * kk;
        const float* Hp = base + (size_t)Q.hrow[l] * 64 + rho * 16 + 4 * kk;
        te = ev_t[e];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) za[mo] = mo < ot ? __builtin_nontemporal_load((gq4)(Zp + 256 * mo)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) hb[mi] = mi < it ? __builtin_nontemporal_load((gq4)(Hp + 256 * mi)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto mac_unit = [&](const f32x4 (&za)[4], const f32x4 (&hb)[4], float te) {
        const float b4 = rho == 0 ? (G.time_dep ? te : 0.f) : (rho == 1 ? 1.f : 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) {
                if (mo < ot) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) if (mi < it) acc[mo][mi] = mfma16(za[mo][j], hb[mi][j], acc[mo][mi]);
                    acc[mo][4] = mfma16(za[mo][j], b4, acc[mo][4]);
                }
            }
        }
    };
    {
        f32x4 zA[4], hA[4], zB[4], hB[4];
        float tA = 0.f, tB = 0.f;
        int u = u_lo + wave;
        if (u < u_hi) load_unit(u, zA, hA, tA);
        while (u < u_hi) {
            const bool nb = u + kCW < u_hi;
            if (nb) load_unit(u + kCW, zB, hB, tB);
            mac_unit(zA, hA, tA);
            u += kCW;
            if (u >= u_hi) break;
            const bool na = u + kCW < u_hi;
            if (na) load_unit(u + kCW, zA, hA, tA);
            mac_unit(zB, hB, tB);
            u += kCW;
        }
    }
    // fixed-order sum over the 4 waves, then scatter into the Flux.destructure order of this chunk's partial vector
    for (int w = 0; w < kCW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo)
#pragma unroll
                for (int mi = 0; mi < 5; ++mi) {
                    f32x4* dst = (f32x4*)(ACC + ((mo * 5 + mi) * 64 + lane) * 4);
                    if (w == 0) *dst = acc[mo][mi]; else *dst += acc[mo][mi];
                }
        }
        __syncthreads();
    }
    float* dstp = wslab + (size_t)chunk * P_total + G.poff[l];
    for (int idx = tid; idx < 20 * 256; idx += 64 * kCW) {
        const int i4 = idx & 3, ln = (idx >> 2) & 63, tl = idx >> 8, mo = tl / 5, mi = tl - 5 * mo;
        const int o = 16 * mo + 4 * (ln >> 4) + i4, j = ln & 15;
        if (o >= out) continue;
        const float v = ACC[idx];
        if (mi < 4) { const int i = 16 * mi + j; if (i < in) dstp[(size_t)i * out + o] = v; }
        else if (j == 0) { if (G.time_dep) dstp[(size_t)in * out + o] = v; }
        else if (j == 1) dstp[(size_t)(in + G.time_dep) * out + o] = v;
    }
}

}  // namespace rnde
