// rnde_bstage_sweep.h -- the stage engine's reverse sweep over a RANGE of attempted steps as ONE launch (headline geometry).
//
// rnde_bstage_attempt_kernel (rnde_bstage_persist.h) reverses one attempted step per launch; what it pays per attempt is the boundary the
// scalar chain needs (dt-bar of attempt n + 1 is a sum over all workgroups): launch ramp, 64 VGPRs of weights per lane, the running
// cotangents (uprev-bar, k1-bar) through HBM, per-attempt arguments from the host -- ~3-4 us of a 30 us attempt, 39 times per step.
// Here the loop over the attempts n_hi, n_hi - 1, .. n_lo runs inside the kernel (what Tracker.gradient does over the taped solve,
// reference experiments/mnist_node.jl:229-232 with src/models/neural_ode.jl:131-137):
//   * the weight slices, the running cotangents U = uprev-bar and K1 = k1-bar and the scalar state (BState) stay in registers;
//   * the three per-workgroup partials {S, tau, sum_j c_j tau_j} of an attempt meet through agent-scope granules (solve_meet,
//     rnde_stage_solve.h) and are summed in the order finish_attempt_scalars_from sums them: bit-identical cotangents;
//   * StepMeta comes from the device copy the forward controller wrote, the host-side per-attempt scalars (cotangent coefficients of the
//     stiffness estimate, pow(qold, beta2), the callback cotangent) from one small array uploaded per reverse pass.
// A sweep may be cut into segments (the host does so to start the weight-gradient GEMMs of the attempts reversed first on the CUs the sweep
// leaves idle): a segment that does not begin at the last attempt reads U, K1, BState and the partials of attempt n_hi + 1 from memory,
// every segment leaves them there for its successor (the next segment, or the kernels that reverse the initial-step rule).
//   * the tape is PREFETCHED: an attempt reads 15 arrays of its record (uprev, unew, k1..k7, h2..h7: 105 KB per workgroup, cold in HBM -- at
//     ~11 B/cycle and CU that is ~6 us of a 30 us attempt when START requests them all at once, as the per-attempt kernel must).  Every wave
//     owns 17 one-KiB LDS slots (its own 16 bytes per lane of each array, brought in by `global_load_lds`); a slot is dead after the
//     stage that reads it last, and is refilled at once with the NEXT attempt's slice of the same array -- behind the stage's hand-off
//     poll, never in front of it (a wave's vector-memory operations return in order) -- so START finds its operands in LDS.
// No saveat (the dense-output cotangents stay on the per-attempt kernels).  All workgroups resident at once; bounded spins; a time-out
// raises the abort word (the host reports the reverse pass as failed: the tape is consumed, as with the per-attempt kernels).
#pragma once
#include "../../../regneuralde.jl_amd/csrc/rnde_bstage_persist.h"
#include "../../../regneuralde.jl_amd/csrc/rnde_stage_solve.h"

namespace rnde {

struct SweepArgs { float c1, c2, svb, pad; double qo, pad2; };   // per attempt: eigen_est cotangent coefficients, callback cotangent, pow(qold_in, beta2)

#ifndef RNDE_SWEEP_PREFETCH
#define RNDE_SWEEP_PREFETCH 1
#endif
#ifdef RNDE_DIAG_SWEEP      // (diagnostic build of this translation unit only: tools/diag_sweep.sh) cycle stamps of workgroup 0, thread 0, in the attempt n_hi - 3
#define WSTAMP_(i) do { if (wg == 0 && tid == 0 && n == n_hi - 3) ((unsigned long long*)(args + 40))[i] = clock64(); } while (0)
#else
#define WSTAMP_(i) do { } while (0)
#endif

template <int ACT2>
__global__ __launch_bounds__(64 * 7) void rnde_bstage_sweep_kernel(const BStageParams Q, const int n_hi, const int n_lo, const SweepArgs* __restrict__ args,
                                                                   const PersistSync Y, const SolveSync Z) {
#pragma clang fp contract(off)
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    constexpr int gWT = 7, gHT = 7, gKHb = 7, gR = 7, gD = 784, gH = 100;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KZ = 16 * gKHb + 4, KG = 16 * gWT + 4;
    float* ZL = smem;
    float* GL = ZL + kSCB * KZ;
    float* RED = GL + kSCB * KG;         // [32]; RED[24]: "a wave of this workgroup gave up"
    double* SUMS = (double*)(RED + 32);  // [4]: the three cross-workgroup sums of the meeting, [3] != 0: the meeting failed
    // tape slots [17][7 waves][256 floats]: 0 uprev, 1 unew, 2 k7, then for the stages j = 6..1: 3 + 2 (6 - j) = h_{j+1}, 4 + 2 (6 - j) = k_j;
    // stage 1 has a second pair (15, 16) and the attempts alternate between the two, so that the next attempt's (h_2, k_1) can be requested
    // a whole stage before this attempt's are dead -- nothing is requested right in front of the meeting
    float* HP = RED + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < 7);
    auto slot_of = [&](int sl) -> float* { return HP + (size_t)(sl * 7 + w) * 256; };
    const int rb = (blockIdx.x >> 3) % gR, ct = 8 * ((blockIdx.x >> 3) / gR) + (blockIdx.x & 7);
    if (ct >= Q.C) return;
    const int wg = rb * Q.C + ct;
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    constexpr bool vec = true;
    const bool writer = (wg == 0 && tid == 0);
    const int T = rb * gWT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const long long A = (long long)gD * P.Bpad;
    const RecLayout L{A, (long long)gH * P.Bpad};
    const size_t co = (size_t)gcol * gD;
    if (tid == 0) Y.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;

    // ---- once per launch: weights, the time column of this wave's hidden tile ----
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < gH) w1t_own[i] = Q.p[(size_t)gH * gD + hr];
    }
    f32x4 wB[7], wD[7];
    {
        const f32x4* pB = Q.pwBt + ((size_t)T * gKHb) * 64 + lane;
        const f32x4* pD = Q.pwDt + ((size_t)w * 49 + rb * gWT) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) wD[kb] = pD[(size_t)kb * 64];
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) wB[kb] = pB[(size_t)kb * 64];
    }
    const int own_h0 = 16 * w + 4 * (lane >> 4);
    const int own_zl0 = col * KZ + 16 * w + (lane >> 4);
    const int own_gl0 = col * KG + 16 * w + (lane >> 4);
    const size_t own_zd0 = (size_t)gcol * gH + own_h0;
    if (tid == 0) { RED[24] = 0.f; SUMS[0] = SUMS[1] = SUMS[2] = SUMS[3] = 0.0; }
    __syncthreads();

    auto phase_d = [&](const f32x4& v, unsigned ex) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[own_gl0 + 4 * i] = v[i];
        __syncthreads();
        const size_t tile0 = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
        const float* gbp = GL + col * KG + 4 * (lane >> 4);
        f32x4 bg[7];
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) bg[kb] = *(const f32x4*)(gbp + 16 * kb);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) {
            acc0 = mfma16(wD[kb][0], bg[kb][0], acc0);
            acc1 = mfma16(wD[kb][1], bg[kb][1], acc1);
            acc0 = mfma16(wD[kb][2], bg[kb][2], acc0);
            acc1 = mfma16(wD[kb][3], bg[kb][3], acc1);
        }
        slab_put(Y.tslab, tile0 + w, lane, acc0 + acc1);
    };

    // running cotangents of (uprev, k1) and the scalar state of the attempt reversed last: registers inside the launch, memory at its ends
    f32x4 cU = {0.f, 0.f, 0.f, 0.f}, cK1 = {0.f, 0.f, 0.f, 0.f};
    // BState of attempt n + 1 (tb_pre, dtb_pre, qoldb, t1b, t0b) and the two fields of its StepMeta the scalar chain reads (dt, flags) are
    // carried in LDS (SC, PV): every wave derives the same values, thread 0 writes them, everybody reads them at the next attempt's START
    // (START's own barrier separates the two).  Carried in registers across the loop -- 12 more VGPRs in a kernel at the 256-register limit --
    // this toolchain (ROCm 7.2 hipcc -O3) placed a temporary of the f64 division S / dt in the register that held dtb_pre (found with
    // tools/dbg/sweep_stop.py; visible in the ISA as v_div_fmas / v_div_fixup writing the live pair).
    double* SC = (double*)(RED + 40);    // [5]
    float* PV = RED + 50;                // [2]: dt, flags (as int bits)
    const bool seg_first = (n_hi == Bq.n_att - 1);
    if (!seg_first) {
        cU = ld4(Bq.U + co, r0, gD, true, vec); cK1 = ld4(Bq.K1 + co, r0, gD, true, vec);
        if (tid == 0) {
            const BState bs = Bq.bstate[(n_hi + 1) & 1];
            SC[0] = bs.tb_pre; SC[1] = bs.dtb_pre; SC[2] = bs.qoldb; SC[3] = bs.t1b; SC[4] = bs.t0b;
            PV[0] = P.meta[n_hi + 1].dt; PV[1] = __int_as_float(P.meta[n_hi + 1].flags);
        }
    }
    __syncthreads();

    // request slot `sl` of the attempt whose StepMeta is mm (this wave's 1 KiB of that array)
    auto fill = [&](const StepMeta& mm, int sl, int dst = -1) {
        if (dst < 0) dst = sl;
        const float* Rr = P.arena + (long long)mm.rec * P.rec_stride;
        const float* ups = P.x; const float* k1s = P.f0;
        if (mm.src >= 0) { const float* Rl = P.arena + (long long)mm.src * P.rec_stride; ups = Rl + L.unew(); k1s = Rl + L.k(7); }
        if (sl == 0) dma_unit((const f32x4*)(ups + co + r0), slot_of(dst));
        else if (sl == 1) dma_unit((const f32x4*)(Rr + L.unew() + co + r0), slot_of(dst));
        else if (sl == 2) dma_unit((const f32x4*)(Rr + L.k(7) + co + r0), slot_of(dst));
        else {
            const int j = 6 - (sl - 3) / 2;
            if ((sl - 3) & 1) dma_unit((const f32x4*)((j >= 2 ? Rr + L.k(j) : k1s) + co + r0), slot_of(dst));
            // (every lane takes part, lanes without four hidden units of their own read the column's first four instead, unused: with the request
            //  under a lane condition the compiler splits the neighbouring unconditional requests into two exec-masked halves, and the half
            //  that starts past lane 0 lands in the wrong place -- lanes 16..63 of wave 6, found with the slot check of the diagnostic build)
            else dma_unit((const f32x4*)(Rr + L.h(j + 1) + (own_h0 + 3 < gH ? own_zd0 : (size_t)gcol * gH)), slot_of(dst));
        }
    };
    {   // the launch's first attempt: all 15 slices at once (what START of the per-attempt kernel does)
        const StepMeta m0 = P.meta[n_hi];
#pragma unroll
        for (int sl = 0; sl < 15; ++sl) fill(m0, sl);
    }

    for (int n = n_hi; n >= n_lo; --n) {
        const bool first = (n == Bq.n_att - 1);
        const StepMeta m = P.meta[n];
#if RNDE_SWEEP_PREFETCH
        const bool more = n > n_lo;
#else      // (A/B: every attempt requests its 15 slices itself, at its start)
        const bool more = false;
        if (n != n_hi) {
#pragma unroll
            for (int sl = 0; sl < 13; ++sl) fill(m, sl);
            fill(m, 13, s1h); fill(m, 14, s1k);
        }
#endif
        const int s1h = ((n_hi - n) & 1) ? 15 : 13, s1k = s1h + 1;      // this attempt's stage-1 slots; the next attempt's are the other pair
        const int s1h_nx = 28 - s1h, s1k_nx = s1h_nx + 1;
        const StepMeta mnx = P.meta[more ? n - 1 : n];      // the attempt reversed next: its slices are requested as this one's slots die
        WSTAMP_(0);
        const SweepArgs ar = args[n];
        const float eig_c1 = ar.c1, eig_c2 = ar.c2;
        float* R = P.arena + (long long)m.rec * P.rec_stride;
        const bool accepted = (m.flags & F_ACCEPT) != 0;
        const float dt = m.dt;
        const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
        if (m.src >= 0) { const float* Rl = P.arena + (long long)m.src * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }
        const bool has_eig = (eig_c1 != 0.f || eig_c2 != 0.f);

        float pS[7], pT[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) { pS[i] = 0.f; pT[i] = 0.f; }
        f32x4 utb = {0.f, 0.f, 0.f, 0.f}, unb = {0.f, 0.f, 0.f, 0.f}, upb0 = {0.f, 0.f, 0.f, 0.f};
        f32x4 exk = {0.f, 0.f, 0.f, 0.f}, exg = {0.f, 0.f, 0.f, 0.f}, gbs[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) gbs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ================= BM_START =================
        {
#pragma clang fp contract(off)
            wait_vm<0>();      // this attempt's slices have landed (and the stores of the attempt before are acknowledged)
            WSTAMP_(1);
            f32x4 upv, unv, kq[7];
            upv = *(const f32x4*)(slot_of(0) + 4 * lane);
            if (!upok) upv = (f32x4){0.f, 0.f, 0.f, 0.f};      // (x is the caller's batch: padded columns read as zero, as ld4 does)
            unv = *(const f32x4*)(slot_of(1) + 4 * lane);
            kq[6] = *(const f32x4*)(slot_of(2) + 4 * lane);
#pragma unroll
            for (int j = 6; j >= 2; --j) kq[j - 1] = *(const f32x4*)(slot_of(4 + 2 * (6 - j)) + 4 * lane);
            kq[0] = *(const f32x4*)(slot_of(s1k) + 4 * lane);
            WSTAMP_(2);
#ifdef RNDE_DIAG_SWEEP      // (debug) the slots against direct loads: one bit per array in word 44 behind the stamps
            {
                unsigned bad = 0;
                auto ne = [](const f32x4& a, const f32x4& b) { return a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]; };
                if (ne(upv, ld4(upsrc + co, r0, gD, upok, upvec))) bad |= 1u;
                if (ne(unv, ld4(R + L.unew() + co, r0, gD, true, vec))) bad |= 2u;
                if (ne(kq[0], ld4(k1p + co, r0, gD, true, vec))) bad |= 4u;
#pragma unroll
                for (int q = 2; q <= 7; ++q) if (ne(kq[q - 1], ld4(R + L.k(q) + co, r0, gD, true, vec))) bad |= (4u << (q - 1));
                if (bad) atomicOr((unsigned*)(args + 40) + 2 * 44, bad);
                if (bad) atomicAdd((unsigned*)(args + 40) + 2 * 44 + 1, 1u);
            }
#endif
            double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0;
            double sc_tb, sc_dtb, sc_qoldb, sc_t1b, sc_t0b;
            if (!first) {
                double S3[3];
                if (n == n_hi) {      // segment start: the partials of attempt n + 1 are in memory (written by the previous segment)
                    f32x4 pe[4];
                    bpart_request(Bq, n + 1, lane, pe);
                    double S = 0, tau = 0, ctau = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (lane + 64 * q < Bq.bpart_n) { S += (double)pe[q][0]; tau += (double)pe[q][1]; ctau += (double)pe[q][2]; }
                    S3[0] = wave_sum_d(S); S3[1] = wave_sum_d(tau); S3[2] = wave_sum_d(ctau);
                } else { S3[0] = SUMS[0]; S3[1] = SUMS[1]; S3[2] = SUMS[2]; }
                {      // finish_attempt_scalars_sums (rnde_bwd.h) on the two fields of attempt n + 1's StepMeta it reads
                    const double pv_tb = SC[0], pv_dtb = SC[1], pv_qoldb = SC[2], pv_t1b = SC[3], pv_t0b = SC[4];
                    const float prev_dt = PV[0]; const int prev_flags = __float_as_int(PV[1]);
                    const double dtb = pv_dtb + S3[0] / (double)prev_dt + S3[2];
                    double tbx = pv_tb + S3[1];
                    t1b = pv_t1b; t0b = pv_t0b;
                    if (prev_flags & F_CLAMP) { t1b += dtb; tbx -= dtb; dtpb = 0; } else dtpb = dtb;
                    tb = tbx; qoldb = pv_qoldb;
                }
            }
            float coef;
            {
                const double N = (double)gD * (double)P.Bn;
                double eb = 0, dtb_pre = 0, q11b = 0, qb = 0, qoldb_in = 0;
                if (accepted) {
                    const bool err_term = Bq.reg_kind == 1 || (Bq.reg_kind == 3 && !(m.eest * dt == 0.f));
                    if (err_term) { const double sb = (double)ar.svb; eb += sb * (double)dt; dtb_pre += sb * (double)m.eest; }
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
                    const double qo = ar.qo;
                    q11b += qb / (qo * (double)kGamma);
                    qoldb_in += -(double)kBeta2 * qb * (double)m.q / (double)m.qold_in;
                }
                if (!(m.flags & F_EZERO) && m.eest > 0.f) eb += q11b * (double)kBeta1 * (double)m.q11 / (double)m.eest;
                coef = m.eest > 0.f ? (float)(eb / (N * (double)m.eest)) : 0.f;
                BState b; b.tb_pre = tb; b.dtb_pre = dtb_pre; b.qoldb = qoldb_in; b.t1b = t1b; b.t0b = t0b; b.pad[0] = b.pad[1] = b.pad[2] = 0;
                if (writer) Bq.bstate[n & 1] = b;
                // (all reads of SC / PV above are complete in every wave only after START's barrier below; the writes wait behind it)
                sc_tb = tb; sc_dtb = dtb_pre; sc_qoldb = qoldb_in; sc_t1b = t1b; sc_t0b = t0b;
            }
            WSTAMP_(3);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            float S = 0.f;
            {
                f32x4 acc = tsBt(0) * kq[0], g6 = tsA(5, 0) * kq[0];
#pragma unroll
                for (int s = 1; s < 7; ++s) { acc += tsBt(s) * kq[s]; if (s < 5) g6 += tsA(5, s) * kq[s]; }
                const f32x4 k6 = kq[5], k7 = kq[6];
                f32x4 uin = {0.f, 0.f, 0.f, 0.f}, k1in = {0.f, 0.f, 0.f, 0.f};
                if (accepted) {
                    if (!first) { uin = cU; k1in = cK1; }
                    else uin = ld4(Bq.ubar + co, r0, gD, colok, false);
                }
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
                    upb0[i] = use_new ? 0.f : skb * P.reltol * sgnf(upv[i]);
                }
                const f32x4 w7 = {0.f, 0.f, 0.f, 0.f};
                f32x4 kb7 = dt * (tsBt(6) * utb + w7);
#pragma unroll
                for (int i = 0; i < 4; ++i) S += k7[i] * kb7[i];
                kb7 += k1in;
                if (has_eig) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = colok;
                        const float d1 = k7[i] - k6[i], d2 = unv[i] - (upv[i] + dt * g6[i]);
                        kb7[i] += ok ? eig_c1 * d1 : 0.f;
                        exk[i] = ok ? -eig_c1 * d1 : 0.f;
                        unb[i] += ok ? eig_c2 * d2 : 0.f;
                        exg[i] = ok ? -eig_c2 * d2 : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ACT2 ? kb7[i] * (1.f - k7[i] * k7[i]) : kb7[i];
                st4(R + L.k(7) + co, r0, gD, true, vec, v);
            }
            pS[0] = S;
            WSTAMP_(4);
            phase_d(v, 1u);
            WSTAMP_(5);
            if (tid == 0) { SC[0] = sc_tb; SC[1] = sc_dtb; SC[2] = sc_qoldb; SC[3] = sc_t1b; SC[4] = sc_t0b; PV[0] = m.dt; PV[1] = __int_as_float(m.flags); }
        }

        bool alive = true;
        auto stage = [&](auto jc) {
#pragma clang fp contract(off)
            constexpr int j = decltype(jc)::value;
            if (!alive) return;
            float h_own[4];
            { const f32x4 hq = *(const f32x4*)(slot_of(j == 1 ? s1h : 3 + 2 * (6 - j)) + 4 * lane); h_own[0] = hq[0]; h_own[1] = hq[1]; h_own[2] = hq[2]; h_own[3] = hq[3]; }
            const f32x4 c_ks = *(const f32x4*)(slot_of(j == 1 ? s1k : 4 + 2 * (6 - j)) + 4 * lane);
            float S = 0.f, tau = 0.f;
            constexpr unsigned ex = (unsigned)(7 - j);
            const int buf = slab_buf(ex);
            f32x4 zs = {0.f, 0.f, 0.f, 0.f};
            const bool dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, w, lane, zs);
            WSTAMP_(6 + 5 * (6 - j));
            float* z1dst = R + L.z1(j + 1);
            const size_t tprev0 = (((size_t)slab_buf(ex + 2u) * Q.C + ct) * gR + rb) * gHT;
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            // the slots that died with the previous stage (START for j = 6) take the next attempt's slices: behind this stage's poll, a whole
            // stage of arithmetic in front of the next one
            if (more && !dead) {
                if constexpr (j == 6) { fill(mnx, 0); fill(mnx, 1); fill(mnx, 2); }
                else { fill(mnx, 3 + 2 * (6 - (j + 1))); fill(mnx, 4 + 2 * (6 - (j + 1))); }
                if constexpr (j == 1) { fill(mnx, 13, s1h_nx); fill(mnx, 14, s1k_nx); }
            }
            const bool unit = own_h0 + 3 < gH, trow = own_h0 == gH;
            f32x4 zv;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float hv = h_own[i]; const float z = zs[i] * (1.f - hv * hv); zv[i] = unit ? z : 0.f; }
            if (rb == 0) {
                if (unit) *(f32x4*)(z1dst + own_zd0) = zv;
#pragma unroll
                for (int i = 0; i < 4; ++i) tau += unit ? w1t_own[i] * zv[i] : ((i == 0 && trow) ? zs[0] : 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) ZL[own_zl0 + 4 * i] = zv[i];
            if (dead && lane == 0) RED[24] = 1.f;
            __syncthreads();
            if (RED[24] != 0.f) { alive = false; return; }
            WSTAMP_(7 + 5 * (6 - j));
            // ---- phase B ----
            f32x4 gb;
            {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* zb = ZL + col * KZ + 4 * (lane >> 4);
                f32x4 bf[7];
#pragma unroll
                for (int kb = 0; kb < 7; ++kb) bf[kb] = *(const f32x4*)(zb + 16 * kb);
#pragma unroll
                for (int kb = 0; kb < 7; ++kb) {
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    if (16 * kb + 4 < gH) acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    if (16 * kb + 8 < gH) acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    if (16 * kb + 12 < gH) acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
                gb = acc0 + acc1;
            }
            WSTAMP_(8 + 5 * (6 - j));
            // ---- phase C ----
            if constexpr (j > 1) slab_clears_done();
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            {
                if (has_eig && j == 5) gb += exg;
                gbs[j - 1] = gb;
                if (j == 6) unb += gb;
                constexpr int jn = j - 1;
                f32x4 kbar = tsA(6, jn) * unb + tsBt(jn) * utb;
#pragma unroll
                for (int s = 1; s <= 5; ++s) {
                    if (s > jn) kbar += tsA(s, jn) * gbs[s - 1];
                }
                kbar = dt * kbar;
                if constexpr (jn >= 1) {
                    const f32x4 ks = c_ks;
#pragma unroll
                    for (int i = 0; i < 4; ++i) S += ks[i] * kbar[i];
                    if (has_eig && j == 6) kbar += exk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ACT2 ? kbar[i] * (1.f - ks[i] * ks[i]) : kbar[i];
                    st4(R + L.k(jn + 1) + co, r0, gD, true, vec, v);
                } else {
                    const f32x4 k1v = c_ks;
#pragma unroll
                    for (int i = 0; i < 4; ++i) S += k1v[i] * kbar[i];
                    f32x4 uo = upb0 + unb;
#pragma unroll
                    for (int s = 1; s <= 5; ++s) uo += gbs[s - 1];
                    f32x4 ko = kbar;
                    if (!accepted) {
                        uo += first ? ld4(Bq.ubar + co, r0, gD, colok, false) : cU;
                        if (!first) ko += cK1;
                    }
                    cU = uo; cK1 = ko;
                    if (n == n_lo) { st4(Bq.U + co, r0, gD, true, vec, uo); st4(Bq.K1 + co, r0, gD, true, vec, ko); }
                }
            }
            if (!colok) tau = 0.f;
            pS[7 - j] = S; pT[7 - j] = tau;
            WSTAMP_(9 + 5 * (6 - j));
            if constexpr (j > 1) phase_d(v, ex + 1u);
            WSTAMP_(10 + 5 * (6 - j));
        };
        stage(std::integral_constant<int, 6>{});
        stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 4>{});
        stage(std::integral_constant<int, 3>{});
        stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 1>{});
        if (!alive) return;

        WSTAMP_(36);
#ifdef RNDE_DIAG_SWEEP      // arrival of every workgroup at the end of the attempt (constant 100 MHz clock: comparable across CUs), and at its START
        if (tid == 0 && n == n_hi - 3) ((unsigned long long*)(args + 100))[wg] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && n == n_hi - 4) ((unsigned long long*)(args + 100))[256 + wg] = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- per-workgroup partials {S, tau, sum_j c_j tau_j}: same reduction order as the per-attempt kernel ----
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float a = wave_sum_f(pS[i]), b = wave_sum_f(pT[i]);
            if (lane == 0) { GL[(i * 3 + 0) * 8 + w] = a; GL[(i * 3 + 1) * 8 + w] = b; GL[(i * 3 + 2) * 8 + w] = 0.f; }
        }
        __syncthreads();
        WSTAMP_(37);
        if (w == 0) {
            // the per-attempt kernel's thread 0 forms these 21 sums of 7 one after the other (147 dependent LDS reads: off its critical path, on
            // this kernel's); here lane l < 21 forms sum l, the cross-stage sums read them back with readlane -- same additions in the same order
            float part = 0.f;
            {
                const int l21 = lane < 21 ? lane : 20;
                for (int q = 0; q < gWT; ++q) part += GL[l21 * 8 + q];
            }
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), i * 3 + 0));
                const float ta = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), i * 3 + 1));
                const float xa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), i * 3 + 2));
                if (i == 0) { o0 = sa; o1 = ta; o2 = xa; }
                else { o0 += sa; o1 += ta; o2 += tsC(7 - i) * ta; }
            }
            if (n == n_lo) {      // the launch's last attempt: the partials go to memory for whoever continues (next segment, initial-step kernels)
                if (lane == 0) { float* o = Bq.bpart + ((size_t)(n & 1) * Bq.bpart_n + wg) * 4; o[0] = o0; o[1] = o1; o[2] = o2; o[3] = 0.f; }
            } else {
                const float mine[3] = {o0, o1, o2};
                double o[3];
                const bool ok = solve_meet(Z, Y, n, Bq.bpart_n, wg, 3, mine, o, lane);
                if (lane == 0) { SUMS[0] = o[0]; SUMS[1] = o[1]; SUMS[2] = o[2]; if (!ok) SUMS[3] = 1.0; }
            }
        }
        WSTAMP_(38);
        __syncthreads();
        WSTAMP_(39);
        if (SUMS[3] != 0.0) return;
        slab_clears_done();      // the clears of this attempt's last stage are acknowledged before the next attempt's first put
    }
}

}  // namespace rnde
