// rnde_fwd.h -- what every forward engine shares: tile loads / stores and the step-size controller (initial-step rule, PI controller).
// (The round-1 column-owner step kernels that used to live here: tools/experiments/column_owner/.)
//
// Reference behaviour being replaced (all of it third-party, see include/rnde.h and SURVEY.md 8a/8c):
//   the body of `solve(prob, Tsit5(); ...)` called from reference src/models/neural_ode.jl:64-70,
//   :131-137 with the dynamics of experiments/mnist_node.jl:41-54 / src/models/basic.jl:16-23.
#pragma once
#include "rnde_device.h"

namespace rnde {

__device__ __forceinline__ f32x4 ld_tile(const float* __restrict__ colbase, int row0, int D, bool ok, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok && row0 < D) {
        if (vec) v = *(const f32x4*)(colbase + row0);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (row0 + i < D) v[i] = colbase[row0 + i];
        }
    }
    return v;
}
__device__ __forceinline__ void st_tile(float* __restrict__ colbase, int row0, int D, bool ok, bool vec, f32x4 v) {
    if (ok && row0 < D) {
        if (vec) *(f32x4*)(colbase + row0) = v;
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (row0 + i < D) colbase[row0 + i] = v[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Controller: state before attempt n, derived redundantly (and identically) by every wave.
// SURVEY.md B.1 (n == 0: finish the initial-step heuristic) and B.4 (PI controller, accept/reject).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void finalize_state(const StepParams& P, StepState& S) {
    if (S.done) return;
    if (!(S.t < P.t1)) { S.done = 1; return; }
    if (S.n_att >= P.max_attempts) { S.done = 1; S.status = 2; return; }
    const float dt = (P.t1 - S.t < S.dtp) ? (P.t1 - S.t) : S.dtp;
    if (dt != dt) { S.done = 1; S.status = 4; }
    else if (!(dt > kDtMin)) { S.done = 1; S.status = 3; }
}

// Large batches: at B = 4096 every one of the 896 workgroups of an attempt launch summed all 1,792 x {1, 3} error partials of the previous attempt in its
// prologue -- seven dependent blocks of cold loads in front of the controller (7 us of a 157 us attempt, round 5's scratch build).  One wave forms the sums
// once, behind the launch that wrote the partials, with sum_partials itself (bit-identical); the prologue reads three doubles (advance_state_t's `sums`).
static __global__ __launch_bounds__(64) void rnde_epart_reduce_kernel(const StepParams P, int m, double* __restrict__ out) {
    const int lane = threadIdx.x;
    const float* ep = P.errpart + (size_t)(m & 1) * 3 * P.nwg;
    const double s0 = sum_partials(ep, P.nwg, lane);
    double s1 = 0, s2 = 0;
    if (P.reg_kind >= 2) { s1 = sum_partials(ep + P.nwg, P.nwg, lane); s2 = sum_partials(ep + 2 * P.nwg, P.nwg, lane); }
    if (lane == 0) { double* o = out + 4 * (m & 1); o[0] = s0; o[1] = s1; o[2] = s2; o[3] = 0; }
}

// pre: the error partials of attempt n - 1 (first 256-entry block, this lane's four), requested by the caller ahead of this call --
// the controller state and the partials are two cold loads that do not depend on each other
// prev: the controller state of attempt n - 1 (P.ctl[(n - 1) & 1]), likewise loaded by the caller ahead of the call (both used when PRE and n > 0)
template <bool PRE>
// sums (optional): the three cross-workgroup sums of attempt n - 1 {r^2, (k7-k6)^2, (unew-g6)^2} already formed by the caller (a kernel that
// runs several attempts meets in memory instead of at a kernel boundary: rnde_chainmw.h MW_SOLVE) -- in the order sum_partials would
// qold_pow (optional): powf(prev.qold, beta2) evaluated by the caller ahead of time (it does not depend on the attempt's error norm)
__device__ __forceinline__ StepState advance_state_t(const StepParams& P, int n, int lane, bool writer, StepState* out, const float (&pre)[4],
                                                     const StepState& prev, const double* sums = nullptr, const float* qold_pow = nullptr) {
    StepState S;
    const double N = (double)P.D * (double)P.Bn;
    if (n == 0) {
        S.last_eest = 1.f; S.live = -1; S.done = 0; S.status = 0; S.n_att = 0; S.n_acc = 0; S.qold = kQoldInit;
        S.pad[0] = S.pad[1] = 0;
        S.next_save = (P.nsave > 0 && P.sv_t[0] == P.t0) ? 1 : 0;   // save_start: t0 itself is a save time
        if (P.forced) {
            S.t = P.forced_t; S.dtp = P.forced_dt;
        } else {
            const float d1 = P.initrec->d1, dt0 = P.initrec->dt0, dtmax = P.t1 - P.t0;
            const double s2 = sum_partials(P.initpart + 2 * P.nwg, P.nwg, lane);
            const float d2 = (float)sqrt(s2 / N) / dt0;
            const float m = d1 > d2 ? d1 : d2;
            float dt1; int c1 = 0;
            if (m <= 1e-15f) { dt1 = fmaxf(1e-6f, dt0 * 1e-3f); c1 = 1; }
            else dt1 = (float)pow(10.0, (double)(-(2.f + log10f(m)) / P.rk_order));
            float dt = 100.f * dt0; int sel = 0;
            if (dt1 < dt) { dt = dt1; sel = 1; }
            if (dtmax < dt) { dt = dtmax; sel = 2; }
            if (writer) { P.initrec->d2 = d2; P.initrec->dt1 = dt1; P.initrec->dt = dt; P.initrec->sel = sel;
                          P.initrec->dt1_const = c1; P.initrec->max_is_d2 = (d2 >= d1); }
            S.t = P.t0; S.dtp = P.replay ? P.replay[0] : dt;
            finalize_state(P, S);
        }
        if (writer) *out = S;
        return S;
    }
    const StepState p = PRE ? prev : P.ctl[(n - 1) & 1];
    if (p.done) { if (writer) *out = p; return p; }
    S = p;
    const bool clamped = !P.forced && (P.t1 - p.t < p.dtp);
    const float dt = clamped ? (P.t1 - p.t) : p.dtp;
    const float* ep = P.errpart + (size_t)((n - 1) & 1) * 3 * P.nwg;   // [parity][{r^2, (k7-k6)^2, (unew-g6)^2}][workgroup]
    const double ss = sums ? sums[0] : sum_partials(ep, P.nwg, lane, PRE ? &pre : nullptr);
    const float eest = (float)sqrt(ss / N);
    float eig = 0.f, en1 = 0.f, en2 = 0.f;
    if (P.reg_kind >= 2) {   // stiffness estimate of the composite algorithm AutoTsit5(Tsit5())
        en1 = (float)sqrt(sums ? sums[1] : sum_partials(ep + P.nwg, P.nwg, lane));
        en2 = (float)sqrt(sums ? sums[2] : sum_partials(ep + 2 * P.nwg, P.nwg, lane));
        eig = en1 / en2;
    }
    const int rec = P.tape ? (n - 1) : (p.live == 0 ? 1 : 0);
    int flags = clamped ? F_CLAMP : 0;
    float q, q11 = 0.f, rej_m = 0.f;
    S.n_att = p.n_att + 1;
    S.last_eest = eest;
    if (!(eest == eest) || isinf(eest)) { S.done = 1; S.status = 4; q = 1.f; }
    else {
        if (eest == 0.f) { q = 1.f / kQmax; flags |= F_EZERO | F_QCLAMP; }
        else {
            q11 = powf(eest, P.beta1);
            q = q11 / (qold_pow ? *qold_pow : powf(p.qold, P.beta2));
            const float qg = q / kGamma, lo = 1.f / kQmax, hi = 1.f / kQmin;
            if (qg < lo) { q = lo; flags |= F_QCLAMP; }
            else if (qg > hi) { q = hi; flags |= F_QCLAMP; }
            else q = qg;
        }
        const float dtmax = P.t1 - P.t0;
        if (P.replay ? (P.replay[2 * (n - 1) + 1] != 0.f) : (eest <= 1.f)) {
            flags |= F_ACCEPT;
            S.qold = eest > kQoldInit ? eest : kQoldInit;
            float dtnew = dt / q;
            if (!P.forced && dtmax < dtnew) { dtnew = dtmax; flags |= F_DTMAXCLAMP; }
            S.t = p.t + dt; S.dtp = dtnew; S.live = rec; S.n_acc = p.n_acc + 1;
            while (S.next_save < P.nsave && P.sv_t[S.next_save] <= S.t) ++S.next_save;   // save times inside (t, t + dt]
        } else {
            rej_m = 1.f / kQmin;
            const float m2 = q11 / kGamma;
            if (m2 < rej_m) { rej_m = m2; flags |= F_REJQ11; }
            float dtp = dt / rej_m;
            if (!P.forced && dtmax < dtp) dtp = dtmax;
            S.dtp = dtp;
        }
        if (P.replay) {   // the given sequence decides what comes next and where the solve ends
            if (n < P.n_replay) S.dtp = P.replay[2 * n]; else S.done = 1;
        }
        if (P.forced) S.done = 1; else finalize_state(P, S);
    }
    if (writer) {
        *out = S;
        StepMeta M;
        M.t = p.t; M.dt = dt; M.dtp_in = p.dtp; M.eest = eest; M.q11 = q11; M.q = q; M.qold_in = p.qold; M.rej_m = rej_m;
        M.flags = flags; M.src = p.live; M.rec = rec; M.eigen = eig; M.n1 = en1; M.n2 = en2; M.pad[0] = M.pad[1] = 0;
        P.meta[n - 1] = M;
    }
    return S;
}
__device__ __forceinline__ StepState advance_state(const StepParams& P, int n, int lane, bool writer, StepState* out) {
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    const StepState nop{};
    return advance_state_t<false>(P, n, lane, writer, out, none, nop);
}

}  // namespace rnde
