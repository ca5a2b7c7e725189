// rnde_bchainmw.h -- reverse pass of the multi-wave chain kernels (rnde_chainmw.h): discretise-then-optimise through the taped
// Tsit5 attempts, same scalar chain (t, dt, EEst, PI controller, initial-step heuristic) as rnde_bchain.h / rnde_bwd.h.
//
// One workgroup of four waves per 16 batch columns, as in the forward.  A J_f^T product is a chain of transposed Dense layers:
// wave w owns INPUT-feature tile w of each layer; the pre-activation cotangent z_l sits in LDS as [feature][16 columns]
// (B operand), the product comes out in the MFMA's D layout and is multiplied there by the activation derivative of the
// layer below -- whose output the FORWARD pass taped in the slab, so nothing is recomputed (the one-wave kernels run the
// chain forward again: 16 layer products per evaluation; here 8) and every taped value is requested before the chain starts.
// z_l goes to LDS (next layer's operand) and to the slab (parameter-gradient kernel, rnde_chain_wgrad_kernel) in one step.
#pragma once
#include "rnde_chainmw.h"
#include "rnde_bchain.h"

namespace rnde {

struct BMwParams {
    BwdParams B;            // B.U / B.K1 / B.UB1 are fragment-order arrays
    MwGeo G;
    RkTab rk;
    const float* tab;
    float* slab;            // activations from the forward (H rows), pre-activation cotangents written here (Z rows)
    long long ev_stride;
    int ntiles;
    const float* sv_t; const float* sv_ubar; int nsave;
    // SWEEP (the whole reverse sweep in one launch): per-attempt arguments as device arrays, and the meeting of rnde_chainmw.h
    const int* sv_lo; const int* sv_hi; const float* eig_c;     // [n_att], [n_att], [n_att][2]
    unsigned long long* xch; unsigned* xcc; unsigned* abort_word; unsigned epoch; int xch_global; int xcd_slot;
};

// J_f^T product at a taped evaluation.  kout = f's value (element-wise), kbar its cotangent; returns gbar (element-wise) and adds
// this lane's share of the time cotangent to tau.  ZA, ZB: 64 x 16 LDS buffers.  sl: slab base of this (evaluation, tile).
template <int NR>
__device__ __forceinline__ void mw_fbwd(const MwGeo& G, const float* FRt, const float* TV, float* ZA, float* ZB, float* __restrict__ sl,
                                        const float (&gin)[NR], const float (&kout)[NR], const float (&kbar)[NR], float (&gb)[NR], float& tau,
                                        int tid, int wave, int lane) {
    const int Lr = G.n_layers, g = lane >> 4, col = lane & 15;
    // taped outputs of layers 0 .. L-2 (= inputs of layers 1 .. L-1) in this wave's D layout: all requested now
    f32x4 oo[kCMaxL];
#pragma unroll
    for (int l = 1; l < kCMaxL; ++l) {
        oo[l] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (l < Lr && wave < G.mt[l] && G.act[l - 1] != 0) {
            const float* hp = sl + (size_t)G.hrow[l] * 64 + (16 * wave + 4 * g) * 16 + col;
            oo[l] = (f32x4){hp[0], hp[16], hp[32], hp[48]};
        }
    }
    float tl = 0.f;
    // z of the last layer, element-wise
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = tid + 256 * r;
        float v = kbar[r];
        if (G.act[Lr - 1] != 0) v *= (1.f - kout[r] * kout[r]);
        if (e < 16 * 16 * G.mt[Lr]) { ZA[e] = v; sl[(size_t)G.zrow[Lr - 1] * 64 + e] = v; }
        if (G.time_dep) tl = fmaf(v, TV[(Lr - 1) * 64 + (e >> 4)], tl);
    }
    __syncthreads();
    float* Zc = ZA; float* Zn = ZB;
#pragma unroll
    for (int l = kCMaxL - 1; l >= 0; --l) {
        if (l < Lr) {
            const int mtin = G.mt[l], mtout = G.mt[l + 1];
            if (wave < mtin) {
                const int mi = wave;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* fr = FRt + ((size_t)G.toff[l] + (size_t)mi * mtout * 4) * 64 + lane;
                const float* zb = Zc + lane;
                float a[4], b[4], a2[4], b2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[j] = fr[j * 64]; b[j] = zb[j * 64]; }
#pragma unroll
                for (int mo = 0; mo < 4; ++mo) {
                    if (mo < mtout) {
                        if (mo + 1 < mtout) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { a2[j] = fr[((mo + 1) * 4 + j) * 64]; b2[j] = zb[((mo + 1) * 4 + j) * 64]; }
                        }
                        acc0 = mfma16(a[0], b[0], acc0); acc1 = mfma16(a[1], b[1], acc1);
                        acc0 = mfma16(a[2], b[2], acc0); acc1 = mfma16(a[3], b[3], acc1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { a[j] = a2[j]; b[j] = b2[j]; }
                    }
                }
                f32x4 o = acc0 + acc1;
                float* zp = Zn + (16 * mi + 4 * g) * 16 + col;
                if (l > 0) {     // z_{l-1} = (W_l^T z_l) .* act'_{l-1}(out_{l-1})
                    if (G.act[l - 1] != 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] *= (1.f - oo[l][i] * oo[l][i]);
                    }
                    float* sp = sl + (size_t)G.zrow[l - 1] * 64 + (16 * mi + 4 * g) * 16 + col;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { zp[i * 16] = o[i]; sp[i * 16] = o[i]; }
                    if (G.time_dep) {
                        const f32x4 wt = *(const f32x4*)(TV + (l - 1) * 64 + 16 * mi + 4 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) tl = fmaf(o[i], wt[i], tl);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) zp[i * 16] = o[i];
                }
            }
            __syncthreads();
            float* t_ = Zc; Zc = Zn; Zn = t_;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v = ((tid + 256 * r) >> 4) < 16 * G.mt[0] ? Zc[tid + 256 * r] : 0.f;   // (rows past the last tile were never written)
        if (G.pre_act) { const float a0 = tanh_fast(gin[r]); v *= (1.f - a0 * a0); }
        gb[r] = v;
    }
    __syncthreads();
    tau += tl;
}

// ---- the latent-ODE shape (rnde_chainmw.h: lat_mt) with the transposed fragments REGISTER STATIONARY: wave w holds, for every layer, the
// ---- fragments of input tile w (8 k-steps of the 20-wide layers' cotangents, 16 of the 50-wide ones); the transposed table never goes to LDS.
// ---- Same arithmetic in the same order as mw_fbwd.
struct LatWeightsT { float a[kLatLayers][16]; };
__device__ __forceinline__ void lat_load_t(const MwGeo& G, const float* __restrict__ tab, LatWeightsT& W, int wave, int lane) {
    const float* FRtg = tab + (size_t)G.nfrag_f * 64 + 1024;
#pragma unroll
    for (int l = 0; l < kLatLayers; ++l) {
        const int mtin = lat_mt(l), mtout = lat_mt(l + 1);
        const int mi = wave < mtin ? wave : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) W.a[l][j] = (j < lat_ks(l + 1)) ? FRtg[((size_t)G.toff[l] + (size_t)mi * mtout * 4 + (j < lat_ks(l + 1) ? j : 0)) * 64 + lane] : 0.f;
    }
}
template <int NR>
__device__ __forceinline__ void mw_fbwd_lat(const MwGeo& G, const LatWeightsT& W, float* ZA, float* ZB, float* __restrict__ sl,
                                            const float (&gin)[NR], const float (&kout)[NR], const float (&kbar)[NR], float (&gb)[NR],
                                            int tid, int wave, int lane) {
    const int g = lane >> 4, col = lane & 15;
    f32x4 oo[kLatLayers];
#pragma unroll
    for (int l = 1; l < kLatLayers; ++l) {
        oo[l] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wave < lat_mt(l) && G.act[l - 1] != 0) {
            const float* hp = sl + (size_t)G.hrow[l] * 64 + (16 * wave + 4 * g) * 16 + col;
            oo[l] = (f32x4){hp[0], hp[16], hp[32], hp[48]};
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = tid + 256 * r;
        float v = kbar[r];
        if (G.act[kLatLayers - 1] != 0) v *= (1.f - kout[r] * kout[r]);
        if (e < 16 * 16 * lat_mt(kLatLayers)) { ZA[e] = v; sl[(size_t)G.zrow[kLatLayers - 1] * 64 + e] = v; }
    }
    __syncthreads();
#pragma unroll
    for (int l = kLatLayers - 1; l >= 0; --l) {
        float* Zc = ((kLatLayers - 1 - l) & 1) ? ZB : ZA;
        float* Zn = ((kLatLayers - 1 - l) & 1) ? ZA : ZB;
        const int mtin = lat_mt(l), mtout = lat_mt(l + 1);
        if (wave < mtin) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* zb = Zc + lane;
            float b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j < lat_ks(l + 1)) b[j] = zb[j * 64];
#pragma unroll
            for (int j = 0; j < 16; ++j) {      // (k-steps that hold cotangents of real features only: rnde_chainmw.h lat_ks)
                if (j < lat_ks(l + 1)) { if (j & 1) acc1 = mfma16(W.a[l][j], b[j], acc1); else acc0 = mfma16(W.a[l][j], b[j], acc0); }
            }
            f32x4 o = acc0 + acc1;
            float* zp = Zn + (16 * wave + 4 * g) * 16 + col;
            if (l > 0) {
                if (G.act[l - 1] != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] *= (1.f - oo[l][i] * oo[l][i]);
                }
                float* sp = sl + (size_t)G.zrow[l - 1] * 64 + (16 * wave + 4 * g) * 16 + col;
#pragma unroll
                for (int i = 0; i < 4; ++i) { zp[i * 16] = o[i]; sp[i * 16] = o[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) zp[i * 16] = o[i];
            }
        }
        __syncthreads();
    }
    float* Zc = (kLatLayers & 1) ? ZB : ZA;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float v = ((tid + 256 * r) >> 4) < 16 * lat_mt(0) ? Zc[tid + 256 * r] : 0.f;
        if (G.pre_act) { const float a0 = tanh_fast(gin[r]); v *= (1.f - a0 * a0); }
        gb[r] = v;
    }
    __syncthreads();
}

// SWEEP = 1: the WHOLE reverse sweep in one launch (the mirror of MW_SOLVE in rnde_chainmw.h): the loop over the attempted steps, last to first,
// runs inside the kernel; weights and geometry are loaded once; the only thing the workgroups exchange per attempt -- their three partial
// sums {<k, k-bar>, tau, c-weighted tau} -- goes through mw_exchange3 (the <= 32 workgroups are pinned to one XCD and meet through its L2), and
// every workgroup carries the scalar chain (BState) in registers: the same double-precision arithmetic on the same sums as the
// launch-per-attempt path, bit for bit.  n_arg = the first attempt to reverse (n_att - 1); the per-attempt arguments come from device arrays.
template <int NR, int TAB = 0, int LAT = 0, int SWEEP = 0>
__global__ __launch_bounds__(kMwThreads) void rnde_bchainmw_kernel(const BMwParams Q, const int n_arg, const StepMeta m_arg, const int sv_lo_arg, const int sv_hi_arg,
                                                                   const float eig_c1_arg, const float eig_c2_arg) {
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    const MwGeo& G = Q.G;
    constexpr int NKD = 4 * NR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* TV = smem + 512;                       // [BV | TV | FRt] as laid out in the table
    float* FRt = smem + 1024;
    float* ZA = FRt + (size_t)G.nfrag_t * 64;
    float* ZB = ZA + 1024;
    float* RED = ZB + 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (SWEEP) { if (!Q.xch_global && (int)(blockIdx.x & 7) != Q.xcd_slot) return; }      // (8 x ntiles launched: the ones that work share one XCD; xch_global: all work, see MW_SOLVE)
    const int tile = (SWEEP && !Q.xch_global) ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if constexpr (SWEEP) { if (tid == 0) Q.xcc[tile] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
    const int gcol = tile * 16 + (tid & 15);
    const bool colok = gcol < P.B;
    const bool writer = (tile == 0 && tid == 0);
    float* RED2 = RED + 16;                                   // SWEEP: the meeting's three sums (doubles) and its verdict, for the other waves
    BState bprev{}; StepMeta mprev = m_arg; double xs[3] = {0.0, 0.0, 0.0};
    LatWeightsT LT;
    if constexpr (SWEEP) {   // weights once per sweep (a launch per attempt requests its cold tape arrays in front of them instead, see below)
        if constexpr (LAT) lat_load_t(G, Q.tab, LT, wave, lane);
        else mw_fill_lds(Q.tab + (size_t)G.nfrag_f * 64, smem, (G.nfrag_t >> 2) + 4, wave, lane);
    }
    // SWEEP: the record of the next attempt to reverse is read COLD from HBM; its nine arrays are requested at the end of the attempt before it,
    // in front of the meeting, and its step record (which says where that record lives) at the top of that attempt
    StepMeta mcur = m_arg, mnext = m_arg;
    float pkq[SWEEP ? (TAB == 2 ? kRkSMax : 7) : 1][NR], pupv[NR], punv[NR];
    bool have_pf = false;
    for (int n = n_arg; ; --n) {
    const StepMeta m = mcur;
    if (SWEEP && n > 0) mnext = P.meta[n - 1];
    const int sv_lo = SWEEP ? Q.sv_lo[n] : sv_lo_arg, sv_hi = SWEEP ? Q.sv_hi[n] : sv_hi_arg;
    const float eig_c1 = SWEEP ? Q.eig_c[2 * n] : eig_c1_arg, eig_c2 = SWEEP ? Q.eig_c[2 * n + 1] : eig_c2_arg;
    const bool first = (n == Bq.n_att - 1);
    constexpr int SM = TAB == 2 ? kRkSMax : 7;               // stages the register arrays are sized for
    const int NS = TAB == 2 ? Q.rk.S : 7;                    // stages of the pair (rnde_chainmw.h: RkTab)
    const ChainRec L{(long long)Q.ntiles * NKD * 64, NS};
    const size_t fo = (size_t)tile * NKD * 64 + tid;
    auto feat = [&](int r) { return (tid + 256 * r) >> 4; };
    auto valid = [&](int r) { return colok && feat(r) < P.D; };
    // Everything this launch reads that does not depend on loaded data -- the previous reversed attempt's partials, the nine tape arrays of
    // the error estimate's reverse -- is requested in front of the weights (a wave's loads return in order; behind the weights these were two
    // more serial round trips to cold memory in a 33 us launch)
    f32x4 pe[4];
    if (!SWEEP && !first) bpart_request(Bq, n + 1, lane, pe);
    const float* R = P.arena + (long long)m.rec * P.rec_stride;
    float kq[SM][NR], upv[NR], unv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (SWEEP && have_pf) {
            upv[r] = pupv[r]; unv[r] = punv[r];
#pragma unroll
            for (int j = 0; j < SM; ++j) kq[j][r] = pkq[SWEEP ? j : 0][r];
        } else {
            upv[r] = R[L.upc() + fo + 256 * r];
            unv[r] = R[L.unew() + fo + 256 * r];
            kq[0][r] = R[L.k1c() + fo + 256 * r];
#pragma unroll
            for (int j = 1; j < SM; ++j) kq[j][r] = (TAB != 2 || j < NS) ? R[L.k(j + 1) + fo + 256 * r] : 0.f;
        }
    }
    if constexpr (!SWEEP) {
        if constexpr (LAT) lat_load_t(G, Q.tab, LT, wave, lane);      // (the latent-ODE shape: transposed fragments in registers, no LDS fill)
        else mw_fill_lds(Q.tab + (size_t)G.nfrag_f * 64, smem, (G.nfrag_t >> 2) + 4, wave, lane);
    }
    auto fbwd = [&](float* slp, const float (&gin_)[NR], const float (&kout_)[NR], const float (&kbar_)[NR], float (&gb_)[NR], float& tau_) {
        if constexpr (LAT) mw_fbwd_lat<NR>(G, LT, ZA, ZB, slp, gin_, kout_, kbar_, gb_, tid, wave, lane);
        else mw_fbwd<NR>(G, FRt, TV, ZA, ZB, slp, gin_, kout_, kbar_, gb_, tau_, tid, wave, lane);
    };
    // ---- scalar chain, identical in every wave; same arithmetic as rnde_bchain_kernel ----
    double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0;
    if (!first) {
        if constexpr (SWEEP) finish_attempt_scalars_sums(bprev, mprev, xs[0], xs[1], xs[2], tb, dtpb, qoldb, t1b, t0b);
        else finish_attempt_scalars_from(Bq, n + 1, lane, &pe, tb, dtpb, qoldb, t1b, t0b);
    }
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
            const double qo = pow((double)m.qold_in, (double)P.beta2);
            q11b += qb / (qo * (double)kGamma);
            qoldb_in += -(double)P.beta2 * qb * (double)m.q / (double)m.qold_in;
        }
        if (!(m.flags & F_EZERO) && m.eest > 0.f) eb += q11b * (double)P.beta1 * (double)m.q11 / (double)m.eest;
        coef = m.eest > 0.f ? (float)(eb / (N * (double)m.eest)) : 0.f;
        BState b; b.tb_pre = tb; b.dtb_pre = dtb_pre; b.qoldb = qoldb_in; b.t1b = t1b; b.t0b = t0b; b.pad[0] = b.pad[1] = b.pad[2] = 0;
        if (writer) Bq.bstate[n & 1] = b;
        bprev = b; mprev = m;
    }

    float S = 0.f, tau = 0.f, ctau = 0.f;   // sum_j <k_j, kbar_j>; sum of time cotangents; c_s-weighted (+ extra dt-bar)
    float* sl0 = Q.slab + (size_t)(2 + (NS - 1) * n) * Q.ev_stride + ((size_t)tile * G.RS) * 64;
    const bool sv_mode = Q.nsave > 0;
    float utb[NR], unb[NR], upb[NR], k1v[NR], Wv[7][NR];
    // ---- A: reverse of the error estimate; seeds of unew-bar / uprev-bar ----
    {
#pragma unroll
        for (int r = 0; r < NR; ++r) k1v[r] = kq[0][r];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float acc = rk_bt<TAB>(Q.rk, 0) * kq[0][r];
#pragma unroll
            for (int j = 1; j < SM; ++j) acc += rk_bt<TAB>(Q.rk, j) * kq[j][r];      // (a table's weights are 0 past its last stage)
            float uin = 0.f;
            if (accepted) {
                if (!first) uin = Bq.U[fo + 256 * r];
                else if (!sv_mode) uin = valid(r) ? Bq.ubar[(size_t)gcol * P.D + feat(r)] : 0.f;
            }
            utb[r] = 0.f; unb[r] = uin; upb[r] = 0.f;
            if (valid(r)) {
                const float ut = dt * acc;
                const float au = fabsf(upv[r]), an = fabsf(unv[r]);
                const bool use_new = !(au > an);
                const float sk = P.abstol + (use_new ? an : au) * P.reltol;
                const float rr = ut / sk;
                const float rb = coef * rr;
                const float skb = -rb * rr / sk;
                utb[r] = rb / sk;
                if (use_new) unb[r] += skb * P.reltol * sgnf(unv[r]); else upb[r] = skb * P.reltol * sgnf(upv[r]);
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) Wv[i][r] = 0.f;
        }
        if (sv_hi > sv_lo) {
            // reverse of the dense output u(ts) = uprev + dt sum_i b_i(theta) k_i, theta = (ts - t)/dt  
            const float tnew = m.t + dt;
            for (int idx = sv_lo; idx < sv_hi; ++idx) {
                const float ts = Q.sv_t[idx];
                const bool at_end = (ts == tnew);
                const float th = (ts - m.t) / dt;
                float bw[7], dbw[7];
                rk_dense<TAB>(Q.rk, th, bw);
                rk_dense_deriv<TAB>(Q.rk, th, dbw);
                float dth = 0.f;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float ub = valid(r) ? Q.sv_ubar[((size_t)gcol * Q.nsave + idx) * P.D + feat(r)] : 0.f;
                    if (at_end) unb[r] += ub;
                    else {
                        upb[r] += ub;
                        float dacc = dbw[0] * kq[0][r];
#pragma unroll
                        for (int i = 0; i < 7; ++i) { Wv[i][r] += bw[i] * ub; if (i) dacc += dbw[i] * kq[i][r]; }
                        dth += ub * dt * dacc;
                    }
                }
                if (!at_end) { tau += -dth / dt; ctau += -dth * th / dt; }
            }
        }
    }
    // Rb[i] = cotangent of k_{s-i} (zero-based) where s is the next stage to be reversed (rolled loop, kBwdShift)
    float Rb[SM - 1][NR], gb[NR], exk[NR], exg[NR];
    const bool has_eig = (eig_c1 != 0.f || eig_c2 != 0.f);
    // ---- B: stage 7 (k7 = f(unew, t + dt)) ----
    {
        float k7[NR], unv7[NR], kb7[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (TAB == 2) k7[r] = R[L.k(NS) + fo + 256 * r];   // (the closing stage's value: a run-time index into kq would leave the registers)
            else k7[r] = kq[6][r];                // (k(7) and unew were read for the error estimate's reverse)
            unv7[r] = unv[r];
            if constexpr (TAB == 2) kb7[r] = dt * (rk_bt<TAB>(Q.rk, NS - 1) * utb[r]);
            else kb7[r] = dt * (rk_bt<TAB>(Q.rk, 6) * utb[r] + Wv[6][r]);
            S += k7[r] * kb7[r];
            if (accepted && !first) kb7[r] += Bq.K1[fo + 256 * r];
            exk[r] = 0.f; exg[r] = 0.f;
            if (has_eig) {   // reverse of eigen_est = ||k7-k6|| / ||unew-g6|| (direct terms: they do not scale with dt, so not in S)
                const bool ok = valid(r);
                const float d1 = ok ? k7[r] - kq[5][r] : 0.f, d2 = ok ? unv7[r] - R[L.g(6) + fo + 256 * r] : 0.f;
                kb7[r] += eig_c1 * d1; exk[r] = -eig_c1 * d1;
                unb[r] += eig_c2 * d2; exg[r] = -eig_c2 * d2;
            }
        }
        float t7 = 0.f;
        fbwd(sl0 + (size_t)(NS - 2) * Q.ev_stride, unv7, k7, kb7, gb, t7);
        tau += t7; ctau += t7;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            unb[r] += gb[r];
            if constexpr (TAB == 2) {
#pragma unroll
                for (int i = 0; i < SM - 1; ++i) { const int j = NS - 2 - i; Rb[i][r] = j >= 0 ? dt * (Q.rk.aN[j] * unb[r] + Q.rk.bt[j] * utb[r]) : 0.f; }   // kbar_{NS-2-i}
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) Rb[i][r] = dt * (rk_a7<TAB>(Q.rk, 5 - i) * unb[r] + rk_bt<TAB>(Q.rk, 5 - i) * utb[r] + Wv[5 - i][r]);   // kbar_{5-i}
            }
            upb[r] += unb[r];
        }
    }
    // ---- C: stages 6..2 ----
    // (a stage's tape operands are requested one stage ahead: at the top of its own iteration they were a cold round trip in front of every
    //  fbwd of the rolled loop)
    float ksn[NR], gsn[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if constexpr (TAB == 2) ksn[r] = R[L.k(NS - 1) + fo + 256 * r]; else ksn[r] = kq[5][r];
        gsn[r] = R[L.g(NS - 1) + fo + 256 * r];
    }
#pragma unroll 1
    for (int s = NS - 2; s >= 1; --s) {   // zero-based: k_s = f(g_s, t + c_s dt), taped as k(s+1), g(s+1)
        float ks[NR], gs[NR], kb[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { ks[r] = ksn[r]; gs[r] = gsn[r]; }
        if (s > 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { ksn[r] = R[L.k(s) + fo + 256 * r]; gsn[r] = R[L.g(s) + fo + 256 * r]; }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            kb[r] = Rb[0][r];
            S += ks[r] * kb[r];
            if (TAB != 2 && has_eig && s == 5) kb[r] += exk[r];          // direct cotangent of k6
        }
        float ts_ = 0.f;
        fbwd(sl0 + (size_t)(s - 1) * Q.ev_stride, gs, ks, kb, gb, ts_);
        tau += ts_; ctau += rk_c<TAB>(Q.rk, s) * ts_;
        if (TAB != 2 && has_eig && s == 5) {
#pragma unroll
            for (int r = 0; r < NR; ++r) gb[r] += exg[r];   // direct cotangent of g6
        }
        float cb[SM - 2];
#pragma unroll
        for (int i = 0; i < SM - 2; ++i) cb[i] = dt * rk_bwd<TAB>(Q.rk, s, i);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int i = 0; i < SM - 2; ++i) Rb[i][r] = Rb[i + 1][r] + cb[i] * gb[r];
            upb[r] += gb[r];
        }
    }
    // ---- D: k1 and outputs (Rb[0] is now the cotangent of k1) ----
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        S += k1v[r] * Rb[0][r];
        float uo = upb[r], ko = Rb[0][r];
        if (!accepted) {
            if (!first) { uo += Bq.U[fo + 256 * r]; ko += Bq.K1[fo + 256 * r]; }
            else if (!sv_mode) uo += valid(r) ? Bq.ubar[(size_t)gcol * P.D + feat(r)] : 0.f;
        }
        Bq.U[fo + 256 * r] = uo;
        Bq.K1[fo + 256 * r] = ko;
    }
    // (time cotangents: every lane's share counts -- rows / columns past D, B carry exact zeros in z)
    S = wave_sum_f(S); tau = wave_sum_f(tau); ctau = wave_sum_f(ctau);
    if (lane == 0) { RED[wave] = S; RED[4 + wave] = tau; RED[8 + wave] = ctau; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, ta = 0.f, ca = 0.f;
        for (int w = 0; w < kMwWaves; ++w) { s += RED[w]; ta += RED[4 + w]; ca += RED[8 + w]; }
        float* o = Bq.bpart + ((size_t)(n & 1) * Bq.bpart_n + tile) * 4;
        o[0] = s; o[1] = ta; o[2] = ca; o[3] = 0.f;       // (SWEEP: read by the reverse of the initialisation after attempt 0, as always)
    }
    if constexpr (!SWEEP) break;
    else {
        if (n == 0) break;
        {   // the next attempt's tape arrays: in flight while the workgroups meet
            const float* Rn = P.arena + (long long)mnext.rec * P.rec_stride;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                pupv[r] = Rn[L.upc() + fo + 256 * r];
                punv[r] = Rn[L.unew() + fo + 256 * r];
                pkq[0][r] = Rn[L.k1c() + fo + 256 * r];
#pragma unroll
                for (int j = 1; j < SM; ++j) pkq[j][r] = (TAB != 2 || j < NS) ? Rn[L.k(j + 1) + fo + 256 * r] : 0.f;
            }
            have_pf = true; mcur = mnext;
        }
        // the meeting: wave 0 publishes this tile's three sums and collects everybody's, in the order finish_attempt_scalars_from adds them
        if (wave == 0) {
            float mine[3] = {0.f, 0.f, 0.f};
            for (int w = 0; w < kMwWaves; ++w) { mine[0] += RED[w]; mine[1] += RED[4 + w]; mine[2] += RED[8 + w]; }
            double o[3];
            const bool ok = mw_exchange3(MwMeet{Q.xch, Q.abort_word, Q.epoch, Q.ntiles, Q.xch_global}, n, mine, o, tile, lane);
            if (lane == 0) { ((double*)RED2)[0] = o[0]; ((double*)RED2)[1] = o[1]; ((double*)RED2)[2] = o[2]; RED2[6] = ok ? 1.f : 0.f; }
        }
        __syncthreads();
        if (RED2[6] == 0.f) return;          // (a meeting timed out: abort word raised; the host falls back to one launch per attempt)
        xs[0] = ((const double*)RED2)[0]; xs[1] = ((const double*)RED2)[1]; xs[2] = ((const double*)RED2)[2];
        __syncthreads();                     // (RED / RED2 are rewritten by the next attempt)
    }
    }
}

// Reverse of the initialisation (mirror of rnde_bchain_init_kernel): PHASE 1 = f1 = f(u1, t0 + dt0) of the initial-step
// heuristic (slab evaluation 1), PHASE 2 = f0 = f(u0, t0) (evaluation 0) and x-bar.
template <int NR, int PHASE>
__global__ __launch_bounds__(kMwThreads) void rnde_bchainmw_init_kernel(const BMwParams Q) {
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    const MwGeo& G = Q.G;
    constexpr int NKD = 4 * NR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* TV = smem + 512;
    float* FRt = smem + 1024;
    float* ZA = FRt + (size_t)G.nfrag_t * 64;
    float* ZB = ZA + 1024;
    float* RED = ZB + 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int gcol = tile * 16 + (tid & 15);
    const bool colok = gcol < P.B;
    const bool writer = (tile == 0 && tid == 0);
    const size_t fo = (size_t)tile * NKD * 64 + tid;
    const double N = (double)P.D * (double)P.Bn;
    auto feat = [&](int r) { return (tid + 256 * r) >> 4; };
    auto valid = [&](int r) { return colok && feat(r) < P.D; };
    mw_fill_lds(Q.tab + (size_t)G.nfrag_f * 64, smem, (G.nfrag_t >> 2) + 4, wave, lane);
    const InitRec ir = *P.initrec;
    const float dt0 = ir.dt0;
    float* sl = Q.slab + (size_t)(PHASE == 1 ? 1 : 0) * Q.ev_stride + ((size_t)tile * G.RS) * 64;
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
            const double mb = dtb * (-1.0 / (double)P.rk_order) * (double)ir.dt1 / mm;      // dt1 = 10^(-(2 + log10 m) / order)
            if (ir.max_is_d2) d2b += mb; else d1b += mb;
        } else if (dt0 * 1e-3f > 1e-6f) dt0b += 1e-3 * dtb;
        const double n2 = (double)ir.d2 * (double)dt0, n2b = d2b / (double)dt0;
        dt0b += -d2b * (double)ir.d2 / (double)dt0;
        const double coef_w = n2 > 0 ? n2b / (N * n2) : 0.0;
        if (writer) { IBState b; b.tb = tb; b.t1b = t1b; b.t0b = t0b; b.dt0b = dt0b; b.d1b = d1b; b.d2b = d2b; b.coef_w = coef_w; b.pad = 0; Bq.ibstate[0] = b; }
        const float cw = (float)coef_w;
        float f0v[NR], f1v[NR], u1v[NR], f1b[NR], gb[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float xv = valid(r) ? P.x[(size_t)gcol * P.D + feat(r)] : 0.f;
            f0v[r] = P.f0[fo + 256 * r]; f1v[r] = P.f1[fo + 256 * r]; u1v[r] = P.u1[fo + 256 * r];
            f1b[r] = 0.f;
            if (valid(r)) { const float sk = P.abstol + fabsf(xv) * P.reltol; f1b[r] = cw * ((f1v[r] - f0v[r]) / sk) / sk; }
        }
        mw_fbwd<NR>(G, FRt, TV, ZA, ZB, sl, u1v, f1v, f1b, gb, tau, tid, wave, lane);
#pragma unroll
        for (int r = 0; r < NR; ++r) { Bq.UB1[fo + 256 * r] = gb[r]; dot += gb[r] * f0v[r]; }
        dot = wave_sum_f(dot); tau = wave_sum_f(tau);
        if (lane == 0) { RED[wave] = dot; RED[4 + wave] = tau; }
        __syncthreads();
        if (tid == 0) {
            float s = 0.f, ta = 0.f;
            for (int w = 0; w < kMwWaves; ++w) { s += RED[w]; ta += RED[4 + w]; }
            float* o = Bq.ipart + (size_t)tile * 4;
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
        float xq[NR], f0v[NR], f0b[NR], u0b[NR], gb[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            xq[r] = valid(r) ? P.x[(size_t)gcol * P.D + feat(r)] : 0.f;
            f0v[r] = P.f0[fo + 256 * r];
            const float f1v = P.f1[fo + 256 * r], ub1 = Bq.UB1[fo + 256 * r];
            const float Uv = Bq.U[fo + 256 * r], K1v = Bq.K1[fo + 256 * r];
            f0b[r] = K1v + dt0 * ub1; u0b[r] = Uv + ub1;
            if (valid(r)) {
                const float sk = P.abstol + fabsf(xq[r]) * P.reltol;
                const float w = (f1v - f0v[r]) / sk, v = f0v[r] / sk, z = xq[r] / sk;
                const float wb = cw * w, vb = cv * v, zb = cz * z;
                const float skb = -(wb * w + vb * v + zb * z) / sk;
                f0b[r] += (vb - wb) / sk;
                u0b[r] += zb / sk + skb * P.reltol * sgnf(xq[r]);
                if (Bq.sv_ubar0) u0b[r] += Bq.sv_ubar0[((size_t)gcol * Bq.sv_T) * P.D + feat(r)];
            }
        }
        mw_fbwd<NR>(G, FRt, TV, ZA, ZB, sl, xq, f0v, f0b, gb, tau, tid, wave, lane);
#pragma unroll
        for (int r = 0; r < NR; ++r) if (valid(r)) Bq.xbar[(size_t)gcol * P.D + feat(r)] = u0b[r] + gb[r];
        tau = wave_sum_f(tau);
        if (lane == 0) RED[wave] = tau;
        __syncthreads();
        if (tid == 0) {
            float ta = 0.f;
            for (int w = 0; w < kMwWaves; ++w) ta += RED[w];
            float* o = Bq.ipart + ((size_t)P.nwg + tile) * 4;
            o[0] = ta; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        }
    }
}

}  // namespace rnde
