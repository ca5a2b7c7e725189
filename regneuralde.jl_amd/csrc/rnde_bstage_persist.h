// rnde_bstage_persist.h -- reverse pass of one attempted step of the stage engine as ONE launch (mirror of
// rnde_stage_persist.h): BM_START and the six BM_STAGE launches of rnde_bstage_kernel fused, with the hbar slab handed
// between the row blocks of a column tile through the XCD's L2 (slab_put / slab_poll_sum).
//   * weights (W1x^T rows, [W2x^T; w2t^T] K-slices) are loaded once; utilde-bar, unew-bar, the uprev-bar seed, the six
//     gbar_s, the dense-output weights W_i and the stiffness extras never leave registers (the multi-launch kernels
//     round-trip them through HBM: UTB/UNB/UPB0/GB/SVW/EXK/EXG);
//   * the tape operands of stage j (h_j, k_{j-1}) are requested before waiting for the hand-off of stage j;
//   * what remains a kernel boundary is what the algorithm needs: dt-bar of attempt n+1 is a sum over ALL workgroups.
// Same arithmetic in the same order as rnde_bstage_kernel (both are compiled with contraction off), so cotangents are
// bit-identical (tests/test_gpu_forward.py::test_persistent_attempt_is_bit_identical).
#pragma once
#ifndef RNDE_BSTAGE_HDMA
#define RNDE_BSTAGE_HDMA 1
#endif
#include "rnde_bstage.h"
#include "rnde_stage_persist.h"
#include "rnde_x3.h"

namespace rnde {

#ifdef RNDE_DIAG
#define BSTAMP(i) do { if (Q.B.F.dbg_out && wg == 0 && tid == 0) ((unsigned long long*)Q.B.F.dbg_out)[i] = clock64(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
#ifndef RNDE_BX3_LATE_WB      // where the X3 form requests its weight fragments (us per reversed attempt, B = 512, same box): 0 = both sets first (26.5); 1 = xB behind the
#define RNDE_BX3_LATE_WB 1     // record's and the stages' requests, at the end of START's queue (25.6, shipped); 2 = xD behind the record's as well (26.8: START's own
#endif                         // phase D then waits for it); 3 = xB behind START's put (26.1 against 24.9 for 1 on the later build: the first poll queues behind it)

// FIX = 1: the headline geometry (D = 784, H = 100, 7 waves, 7 row blocks) as compile-time constants, see rnde_stage_attempt_kernel
// X3 = 1 (with FIX; every callback, with saveat (SV = 1) or without): the two
// transposed products of every stage on the matrix cores (rnde_x3.h: exact three-way bf16 split, six v_mfma_f32_16x16x32_bf16 per 32 k-values), as the
// forward solve of matrix mode 1 forms its own.  Not bit-identical to the fp32-input-MFMA form; parity vs the fp64 restatement: tests/test_gpu_x3.py.
// SV: which optional cotangents the instantiation carries -- bit 0 the saveat ones, bit 1 the eigen_est ones.  The fp32 forms carry both (3) behind run-time tests, as since
// round 2; the X3 form is instantiated per combination: 0 is the headline (198 VGPRs, no tests in the stages), with saveat it sits at the 256-register limit
template <int ACT2, int FIX, int X3 = 0, int SV = 3>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_bstage_attempt_kernel(const BStageParams Q, const int n, const StepMeta m, const float eig_c1,
                                                                          const float eig_c2, const int sv_lo, const int sv_hi, const PersistSync Y, const double qo_host,
                                                                          const float svb_n /* = svb_att[n], known to the host: saves a dependent load in the scalar chain */) {
#pragma clang fp contract(off)
    const BwdParams& Bq = Q.B;
    const StepParams& P = Bq.F;
    const int gWT = FIX ? 7 : Q.WT, gHT = FIX ? 7 : Q.HT, gKHb = FIX ? 7 : Q.KHb, gMT = FIX ? 49 : Q.MT, gR = FIX ? 7 : Q.R;
    const int gD = FIX ? 784 : P.D, gH = FIX ? 100 : P.H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KZ = 16 * gKHb + 4, KG = 16 * gWT + 4;
    static_assert(!X3 || FIX, "the X3 form exists for the headline geometry only");
    float* ZL = smem;
    float* GL = ZL + (X3 ? kX3ImageFloats : kSCB * KZ);
    float* RED = GL + (X3 ? kX3ImageFloats : kSCB * KG);         // [32]; RED[24..31]: per-wave "gave up" flags of the hand-off
    unsigned short* ZX = (unsigned short*)ZL;      // X3: operand images [plane][column][kX3K] of bf16
    unsigned short* GX = (unsigned short*)GL;
#if RNDE_BSTAGE_HDMA
    // FIX: the six stages' tape operands (a lane's 16 bytes of h_{j+1} and of k_j) are brought into LDS by START with `global_load_lds`, 1 KiB per
    // wave, stage and array -- a wave's vector-memory operations return in order, so a stage that requests its own (cold) operands in front of
    // its poll makes every poll wait for HBM
    float* HP = RED + 64;                // [6 stages][7 waves][2 arrays][256 floats]
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (FIX) __builtin_assume(w >= 0 && w < 7);
    const int rb = (blockIdx.x >> 3) % gR, ct = 8 * ((blockIdx.x >> 3) / gR) + (blockIdx.x & 7);   // see rnde_stage_persist.h
    if (ct >= Q.C) return;
    const int wg = rb * Q.C + ct;
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (gD & 3) == 0;
    const bool writer = (wg == 0 && tid == 0);
    const int T = rb * gWT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const bool tile_ok = FIX ? true : T < gMT;      // (FIX: 7 x 7 = 49 row tiles, none missing)
    const long long A = (long long)gD * P.Bpad;
    const RecLayout L{A, (long long)gH * P.Bpad};
    const bool first = (n == Bq.n_att - 1);
    const size_t co = (size_t)gcol * gD;
    if (tid == 0) Y.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;

    BSTAMP(0);
    // the cross-workgroup partials of attempt n + 1 are requested before anything else: they come back first, and the double-precision scalar
    // chain that needs them then runs while the weights and the nine tape arrays of this attempt are still streaming in
    f32x4 pe[4];
    if (!first && !Bq.bsum) bpart_request(Bq, n + 1, lane, pe);
    __builtin_amdgcn_sched_barrier(0);
    // (the small loads first, and all four weight base addresses exist -- pinned by the empty asm -- before the first weight load: the register
    //  allocator otherwise builds the later addresses in registers that are destinations of loads in flight, and each such reuse is a full wait
    //  for everything issued so far: the partials above and seven weight loads, ~2 k cycles, in front of the rest of the prologue)
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < gH) w1t_own[i] = Q.p[(size_t)gH * gD + hr];
    }
    typedef const __attribute__((address_space(1))) f32x4* gw4;     // (through the asm the compiler no longer knows the pointers are global)
    unsigned long long aB = (unsigned long long)(Q.pwBt + ((size_t)T * gKHb) * 64 + lane);
    unsigned long long aD = (unsigned long long)(Q.pwDt + ((size_t)w * gMT + rb * gWT) * 64 + lane);
    unsigned long long aB4 = aB + 4 * 1024, aD4 = aD + 4 * 1024;       // (the offset field of a load reaches 4095 bytes)
    asm volatile("" : "+v"(aB), "+v"(aD), "+v"(aB4), "+v"(aD4));
    f32x4 wB[X3 ? 1 : kSMaxHT], wD[X3 ? 1 : kSMaxW];
    x3u4 xB[X3 ? 4 : 1][3], xD[X3 ? 4 : 1][3];
    unsigned long long xb_addr[3] = {0ull, 0ull, 0ull}, xd_addr[3] = {0ull, 0ull, 0ull};
    if constexpr (X3) {
        typedef const __attribute__((address_space(1))) x3u4* gx4;
        unsigned long long bD = (unsigned long long)((const x3u4*)Q.x3Dt + ((size_t)(w * gR + rb) * 4 * 3) * 64 + lane);
        unsigned long long bB = (unsigned long long)((const x3u4*)Q.x3Bt + ((size_t)T * 4 * 3) * 64 + lane);
        unsigned long long bD1 = bD + 4 * 1024, bD2 = bD + 8 * 1024, bB1 = bB + 4 * 1024, bB2 = bB + 8 * 1024;      // (the offset field of a load reaches 4095 bytes)
        asm volatile("" : "+v"(bD), "+v"(bB), "+v"(bD1), "+v"(bD2), "+v"(bB1), "+v"(bB2));
#if RNDE_BX3_LATE_WB < 2
#pragma unroll
        for (int f = 0; f < 12; ++f) xD[f / 3][f % 3] = f < 4 ? ((gx4)bD)[(size_t)f * 64] : (f < 8 ? ((gx4)bD1)[(size_t)(f - 4) * 64] : ((gx4)bD2)[(size_t)(f - 8) * 64]);
#else
        xd_addr[0] = bD; xd_addr[1] = bD1; xd_addr[2] = bD2;      // (xD -- first multiplied in START's own phase D -- behind the record's requests, see there)
#endif
#if !RNDE_BX3_LATE_WB
#pragma unroll
        for (int f = 0; f < 12; ++f) xB[f / 3][f % 3] = f < 4 ? ((gx4)bB)[(size_t)f * 64] : (f < 8 ? ((gx4)bB1)[(size_t)(f - 4) * 64] : ((gx4)bB2)[(size_t)(f - 8) * 64]);
#else
        // (xB -- the fragments of phase B, first multiplied in stage 6 -- are requested at the END of START's queue, behind the record and the stages' operands:
        //  84 KB per workgroup less in front of the arrays START itself waits for)
        xb_addr[0] = bB; xb_addr[1] = bB1; xb_addr[2] = bB2;
#endif
        // k-values 112 .. 135 of every (plane, column) row are written by nobody (they multiply zero weights, but must not hold NaN patterns another kernel
        // left in LDS): zeroed here.  ONLY those -- rows 0 .. 111 are written by the waves' own x3_store4 with no barrier between this loop and START's.
        for (int i = tid; i < 2 * 3 * 16 * 12; i += 64 * 7) ((unsigned*)ZL)[(i / 12) * (kX3K / 2) + 56 + i % 12] = 0u;
    } else {
    // (wD first: the prologue's own phase D needs it; wB is not multiplied before phase B of the first stage and streams in behind)
#pragma unroll
    for (int kb = 0; kb < kSMaxW; ++kb)
        if (kb < gWT && w < gHT && rb * gWT + kb < gMT) wD[kb] = kb < 4 ? ((gw4)aD)[(size_t)kb * 64] : ((gw4)aD4)[(size_t)(kb - 4) * 64];
#pragma unroll
    for (int kb = 0; kb < kSMaxHT; ++kb)
        if (kb < gKHb && tile_ok) {
            if (FIX && kb == 6) {   // only k-steps 96..99 of the last block are multiplied (phase B): the rest of the float4 would be dead registers the
                typedef const __attribute__((address_space(1))) float* gw1;      // moment it is requested, and the allocator's reuse of them a wait for
                wB[kb] = (f32x4){*(gw1)(aB4 + 2 * 1024), 0.f, 0.f, 0.f};       // every load issued before it
            } else wB[kb] = kb < 4 ? ((gw4)aB)[(size_t)kb * 64] : ((gw4)aB4)[(size_t)(kb - 4) * 64];
        }
    }
    BSTAMP(43);
    float* R = P.arena + (long long)m.rec * P.rec_stride;
    BSTAMP(44);
    const float* W1t = Q.p + (size_t)gH * gD;
    const bool accepted = (m.flags & F_ACCEPT) != 0;
    const float dt = m.dt;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (m.src >= 0) { const float* Rl = P.arena + (long long)m.src * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }
    const bool has_eig = (SV & 2) ? (eig_c1 != 0.f || eig_c2 != 0.f) : false;
    const bool has_sv = (SV & 1) ? sv_hi > sv_lo : false;

    // per-stage partials {S, tau, exdt}: index 0 = START, 1..6 = stage j = 6..1 (reduced at the end in launch order)
    float pS[7], pT[7], pX[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { pS[i] = 0.f; pT[i] = 0.f; pX[i] = 0.f; }

    // values that the multi-launch kernels pass through HBM between the launches of an attempt
    f32x4 utb = {0.f, 0.f, 0.f, 0.f}, unb = {0.f, 0.f, 0.f, 0.f}, upb0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 exk = {0.f, 0.f, 0.f, 0.f}, exg = {0.f, 0.f, 0.f, 0.f}, Wv[7], gbs[6];
#pragma unroll
    for (int i = 0; i < 7; ++i) Wv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) gbs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // loop-invariant addressing of this lane's four rows of its own hidden tile / row tile (kperm(16 w + 4 g + i) = 16 w + g + 4 i):
    // the stages are instruction bound between the hand-offs, nothing that does not change is recomputed per stage
    const int own_h0 = 16 * w + 4 * (lane >> 4);                       // first of the four hidden rows
    const int own_zl0 = col * KZ + 16 * w + (lane >> 4);               // ZL slot of row own_h0 (+ 4 i)
    const int own_gl0 = col * KG + 16 * w + (lane >> 4);               // GL slot of row 16 w + 4 g (+ 4 i)
    const size_t own_zd0 = (size_t)gcol * gH + own_h0;                // tape offset of (column, own_h0) in the H x B arrays
    if (tid == 0) RED[24] = 0.f;                                       // "a wave of this workgroup gave up"
    auto phase_d = [&](const f32x4& v, unsigned ex) {
        if constexpr (X3) {
            x3_store4(GX, col, 16 * w + 4 * (lane >> 4), v);
            __syncthreads();
            const size_t tile0x = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
            slab_put(Y.tslab, tile0x + w, lane, x3_tile<4>(xD, GX, lane));
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[own_gl0 + 4 * i] = tile_ok ? v[i] : 0.f;
        __syncthreads();
        const size_t tile0 = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
        const float* gbp = GL + col * KG + 4 * (lane >> 4);
        f32x4 bg[kSMaxW];
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb) if (kb < gWT) bg[kb] = *(const f32x4*)(gbp + 16 * kb);
        if (w < gHT) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < gWT && rb * gWT + kb < gMT) {
                    acc0 = mfma16(wD[kb][0], bg[kb][0], acc0);
                    acc1 = mfma16(wD[kb][1], bg[kb][1], acc1);
                    acc0 = mfma16(wD[kb][2], bg[kb][2], acc0);
                    acc1 = mfma16(wD[kb][3], bg[kb][3], acc1);
                }
            }
            slab_put(Y.tslab, tile0 + w, lane, acc0 + acc1);
        }
        for (int ht = w + gWT; ht < gHT; ht += gWT) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < gWT && rb * gWT + kb < gMT) {
                    const f32x4 a = Q.pwDt[((size_t)ht * gMT + rb * gWT + kb) * 64 + lane];
                    acc0 = mfma16(a[0], bg[kb][0], acc0);
                    acc1 = mfma16(a[1], bg[kb][1], acc1);
                    acc0 = mfma16(a[2], bg[kb][2], acc0);
                    acc1 = mfma16(a[3], bg[kb][3], acc1);
                }
            }
            slab_put(Y.tslab, tile0 + ht, lane, acc0 + acc1);
        }
    };

    // ================= BM_START =================
    {
#pragma clang fp contract(off)
        // the attempt's arrays are requested first: they do not depend on the scalar chain below, whose own loads (partials of
        // attempt n+1, BState) and double-precision arithmetic would otherwise sit in front of them (START: 21 k cycles)
        f32x4 upv = {0.f, 0.f, 0.f, 0.f}, unv = upv, kq[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) kq[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tile_ok) {
            upv = ld4(upsrc + co, r0, gD, upok, upvec);
            unv = ld4(R + L.unew() + co, r0, gD, true, vec);
            kq[0] = ld4(k1p + co, r0, gD, true, vec);
#pragma unroll
            for (int s = 2; s <= 7; ++s) kq[s - 1] = ld4(R + L.k(s) + co, r0, gD, true, vec);
        }
        // X3 form: the running cotangents (written by the previous launch: L2 / Infinity Cache) are requested WITH the record, not behind its arrival --
        // two small loads in front of the scalar chain (round 5 moved them together with the twelve cold DMA requests and lost: those stay where they were)
        f32x4 uin_e = {0.f, 0.f, 0.f, 0.f}, k1in_e = {0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            if (tile_ok && (m.flags & F_ACCEPT) && !first) { uin_e = ld4(Bq.U + co, r0, gD, true, vec); k1in_e = ld4(Bq.K1 + co, r0, gD, true, vec); }
        }
#if RNDE_BX3_LATE_WB >= 2
        if constexpr (X3) {
            typedef const __attribute__((address_space(1))) x3u4* gx4;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 12; ++f) xD[f / 3][f % 3] = ((gx4)xd_addr[f >> 2])[(size_t)(f & 3) * 64];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);   // keep these requests in front of the scalar chain (the scheduler sinks them to their uses otherwise)
        BSTAMP(40);
        double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0;
        if (!first) {
            if (Bq.bsum) {      // (large batches: the three sums were formed once behind the previous launch, rnde_bpart_reduce_kernel -- the same additions in the same order)
                const double* sm = Bq.bsum + 4 * ((n + 1) & 1);
                finish_attempt_scalars_sums(Bq.bstate[(n + 1) & 1], P.meta[n + 1], sm[0], sm[1], sm[2], tb, dtpb, qoldb, t1b, t0b);
            } else finish_attempt_scalars_from(Bq, n + 1, lane, &pe, tb, dtpb, qoldb, t1b, t0b);
        }
        BSTAMP(41);
        float coef;
        {
            const double N = (double)gD * (double)P.Bn;
            double eb = 0, dtb_pre = 0, q11b = 0, qb = 0, qoldb_in = 0;
            if (accepted) {
                const bool err_term = Bq.reg_kind == 1 || (Bq.reg_kind == 3 && !(m.eest * dt == 0.f));
                if (err_term) { const double sb = (double)svb_n; eb += sb * (double)dt; dtb_pre += sb * (double)m.eest; }
                if (Bq.reg_kind == 4 && !(m.eigen == 0.f || m.eigen != m.eigen)) dtb_pre += (double)svb_n * ((double)m.eigen * (double)dt > 0 ? 1.0 : -1.0) * (double)m.eigen;      // |eigen_est * dt|: its dt share (the eigen_est share travels as eig_c1 / eig_c2)
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
        BSTAMP(42);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        float S = 0.f, tau = 0.f, exdt = 0.f;
        if (tile_ok) {
            f32x4 acc = tsBt(0) * kq[0], g6 = tsA(5, 0) * kq[0];
#pragma unroll
            for (int s = 1; s < 7; ++s) { acc += tsBt(s) * kq[s]; if (s < 5) g6 += tsA(5, s) * kq[s]; }
            const f32x4 k6 = kq[5], k7 = kq[6];
            BSTAMP(45);      // (the seven k arrays of the record have arrived: everything before this stamp since 42 is waiting for the tape)
            f32x4 uin = {0.f, 0.f, 0.f, 0.f}, k1in = {0.f, 0.f, 0.f, 0.f};
            const bool sv_mode = Q.nsave > 0;
            if (accepted) {
                if (!first) { if constexpr (X3) { uin = uin_e; k1in = k1in_e; } else { uin = ld4(Bq.U + co, r0, gD, true, vec); k1in = ld4(Bq.K1 + co, r0, gD, true, vec); } }
                else if (!sv_mode) uin = ld4(Bq.ubar + co, r0, gD, colok, false);
            }
#if RNDE_BSTAGE_HDMA
            if constexpr (FIX) {
#pragma unroll
                for (int j = 6; j >= 1; --j) {
                    float* slot = HP + (size_t)(((6 - j) * 7 + w) * 2) * 256;
                    // (measured and dropped in round 6: each stage's h requested one stage ahead, behind the previous stage's poll, instead of all six here --
                    //  27.25 against 26.5 us per reversed attempt: a cold request in the wave's in-order queue costs the NEXT poll more than START saves)
                    if (own_h0 + 3 < gH) dma_unit((const f32x4*)(R + L.h(j + 1) + own_zd0), slot);
                    // k_j: the lane's 16 bytes are in kq[j - 1] already (START read k1..k7 for the error estimate's cotangent): the X3 form parks them in the
                    // slot with an LDS store instead of requesting them a second time -- 42 KB less per workgroup through the CU's 64 B/clk vector-memory
                    // path, which is what START waits on (the fp32 form keeps the request: its results stay bit for bit what the tests pin)
                    if constexpr (X3) *(f32x4*)(slot + 256 + 4 * lane) = kq[j - 1];
                    else dma_unit((const f32x4*)((j >= 2 ? R + L.k(j) : k1p) + co + r0), slot + 256);
                }
            }
#endif
#if RNDE_BX3_LATE_WB == 1
            if constexpr (X3) {
                typedef const __attribute__((address_space(1))) x3u4* gx4;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 12; ++f) xB[f / 3][f % 3] = ((gx4)xb_addr[f >> 2])[(size_t)(f & 3) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ut = dt * acc[i];
                const float au = fabsf(upv[i]), an = fabsf(unv[i]);
                const bool use_new = !(au > an);
                const float sk = P.abstol + (use_new ? an : au) * P.reltol;
                float r, rb_, skb;
                if constexpr (X3) {      // one IEEE division per element instead of three (the other two become products with 1 / sk: each within an ulp of the quotient)
                    const float isk = 1.f / sk;
                    r = ut * isk;
                    rb_ = colok ? coef * r : 0.f;
                    skb = -rb_ * r * isk;
                    utb[i] = rb_ * isk;
                } else {
                    r = ut / sk;
                    rb_ = colok ? coef * r : 0.f;
                    skb = -rb_ * r / sk;
                    utb[i] = rb_ / sk;
                }
                unb[i] = uin[i] + (use_new ? skb * P.reltol * sgnf(unv[i]) : 0.f);
                upb0[i] = use_new ? 0.f : skb * P.reltol * sgnf(upv[i]);
            }
            f32x4 w7 = {0.f, 0.f, 0.f, 0.f};
            if (has_sv) {
                const float tnew = m.t + dt;
                for (int idx = sv_lo; idx < sv_hi; ++idx) {
                    const float ts = Q.sv_t[idx];
                    const f32x4 ub = ld4(Q.sv_ubar + ((size_t)gcol * Q.nsave + idx) * gD, r0, gD, colok, vec);
                    if (ts == tnew) { unb += ub; continue; }
                    const float th = (ts - m.t) / dt;
                    float bw[7], dbw[7];
                    dense_weights(th, bw);
                    dense_weights_deriv(th, dbw);
                    upb0 += ub;
                    f32x4 dacc = dbw[0] * kq[0];
#pragma unroll
                    for (int i = 0; i < 7; ++i) { Wv[i] += bw[i] * ub; if (i) dacc += dbw[i] * kq[i]; }
                    float dth = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dth += ub[i] * dt * dacc[i];
                    tau += -dth / dt;
                    exdt += -dth * th / dt;
                }
                w7 = Wv[6];
            }
            f32x4 kb7 = dt * (tsBt(6) * utb + w7);
#pragma unroll
            for (int i = 0; i < 4; ++i) S += k7[i] * kb7[i];
            kb7 += k1in;
            if (has_eig) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = colok && (r0 + i < gD);
                    const float d1 = k7[i] - k6[i], d2 = unv[i] - (upv[i] + dt * g6[i]);
                    kb7[i] += ok ? eig_c1 * d1 : 0.f;
                    exk[i] = ok ? -eig_c1 * d1 : 0.f;
                    unb[i] += ok ? eig_c2 * d2 : 0.f;
                    exg[i] = ok ? -eig_c2 * d2 : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (FIX || r0 + i < gD) ? (ACT2 ? kb7[i] * (1.f - k7[i] * k7[i]) : kb7[i]) : 0.f;
            st4(R + L.k(7) + co, r0, gD, true, vec, v);
        }
        if (!colok) { tau = 0.f; exdt = 0.f; }
        BSTAMP(1);
        phase_d(v, 1u);
        // the wave sums of this part's partials are formed HERE, behind the put and in front of a poll that has to wait anyway, not in END (where fifteen
        // dependent cross-lane reductions sat at the very end of every reversed attempt); the same function on the same values: the same bits
#if RNDE_BX3_LATE_WB == 3
        if constexpr (X3) {      // xB behind START's put: between the put and a poll that has to wait for six other row blocks anyway
            typedef const __attribute__((address_space(1))) x3u4* gx4;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 12; ++f) xB[f / 3][f % 3] = ((gx4)xb_addr[f >> 2])[(size_t)(f & 3) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        pS[0] = wave_sum_f(S); pT[0] = wave_sum_f(tau); pX[0] = wave_sum_f(exdt);
        BSTAMP(2);
    }

    bool alive = true;
    // ================= BM_STAGE j = 6..1 =================
    auto stage = [&](auto jc) {
#pragma clang fp contract(off)
        constexpr int j = decltype(jc)::value;
        if (!alive) return;
        // tape operands of this stage: independent of the hand-off, so request them first
        float h_own[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FIX) {
#if RNDE_BSTAGE_HDMA
            if (j == 6) wait_vm<0>();      // START's requests have landed (its own stores too: the poll below would wait for those anyway)
            { const f32x4 hq = *(const f32x4*)(HP + (size_t)(((6 - j) * 7 + w) * 2) * 256 + 4 * lane); h_own[0] = hq[0]; h_own[1] = hq[1]; h_own[2] = hq[2]; h_own[3] = hq[3]; }
#else
            if (own_h0 + 3 < gH) { const f32x4 hq = *(const f32x4*)(R + L.h(j + 1) + own_zd0); h_own[0] = hq[0]; h_own[1] = hq[1]; h_own[2] = hq[2]; h_own[3] = hq[3]; }
#endif
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (own_h0 + i < gH) h_own[i] = (R + L.h(j + 1))[own_zd0 + i];
            }
        }
        f32x4 c_ks = {0.f, 0.f, 0.f, 0.f};
#if RNDE_BSTAGE_HDMA
        if constexpr (FIX) c_ks = *(const f32x4*)(HP + (size_t)(((6 - j) * 7 + w) * 2 + 1) * 256 + 4 * lane);
        else
#endif
        if (tile_ok) c_ks = (j >= 2) ? ld4(R + L.k(j) + co, r0, gD, true, vec) : ld4(k1p + co, r0, gD, true, vec);
        float S = 0.f, tau = 0.f;
        // ---- phase A: poll this wave's hidden tile of the R row blocks (the polling load is the data load) ----
        constexpr unsigned ex = (unsigned)(7 - j);          // the exchange this stage consumes (1 = START's put)
        const int buf = slab_buf(ex);
        bool dead = false;
        f32x4 zs = {0.f, 0.f, 0.f, 0.f};
        if (w < gHT) dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, w, lane, zs);
        BSTAMP(3 + 5 * (6 - j));
        const float* hsrc = R + L.h(j + 1);
        float* z1dst = R + L.z1(j + 1);
        // every row block has produced exchange ex, hence consumed ex - 1: this wave's entries of that buffer can be emptied
        const size_t tprev0 = (((size_t)slab_buf(ex + 2u) * Q.C + ct) * gR + rb) * gHT;     // (ex - 1) % 3 == (ex + 2) % 3
        if constexpr (FIX) {
            // one hidden tile per wave; a lane's four rows are four hidden units (all waves but 6, and lanes 0..15 of wave 6), or the
            // t row followed by padding (lanes 16..31 of wave 6), or padding: no per-row branches, one 16-byte tape store
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            const bool unit = own_h0 + 3 < gH, trow = own_h0 == gH;
            f32x4 zv;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float hv = h_own[i]; const float z = zs[i] * (1.f - hv * hv); zv[i] = unit ? z : 0.f; }
            if (rb == w && unit) *(f32x4*)(z1dst + own_zd0) = zv;      // (every row block holds the same z1-bar: row block rb writes hidden tile rb)
            if (rb == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tau += unit ? w1t_own[i] * zv[i] : ((i == 0 && trow) ? zs[0] : 0.f);
            }
            if constexpr (X3) x3_store4(ZX, col, 16 * w + 4 * (lane >> 4), zv);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ZL[own_zl0 + 4 * i] = zv[i];
            }
        } else if (w < gHT) {      // this wave's own hidden tile (addressing precomputed)
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = own_h0 + i;
                float zv = 0.f;
                if (hr < gH) {
                    const float hv = h_own[i];
                    zv = zs[i] * (1.f - hv * hv);
                    if (rb == 0) { z1dst[own_zd0 + i] = zv; tau += w1t_own[i] * zv; }
                } else if (hr == gH) {
                    if (rb == 0) tau += zs[i];
                }
                if (hr < 16 * gKHb) ZL[own_zl0 + 4 * i] = zv;
            }
        }
        for (int ht = w + gWT; ht < gHT; ht += gWT) {      // further hidden tiles (more tiles than waves): the general form
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (!dead) dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, ht, lane, z);
            if (!dead) slab_clear(Y.tslab, tprev0 + ht, lane);
            const int h0 = 16 * ht + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                float zv = 0.f;
                if (hr < gH) {
                    const float hv = hsrc[(size_t)gcol * gH + hr];
                    zv = z[i] * (1.f - hv * hv);
                    if (rb == 0) { z1dst[(size_t)gcol * gH + hr] = zv; tau += W1t[hr] * zv; }
                } else if (hr == gH) {
                    if (rb == 0) tau += z[i];
                }
                if (hr < 16 * gKHb) ZL[col * KZ + kperm(hr)] = zv;
            }
        }
        if (gKHb > gHT) {
            for (int i = tid; i < kSCB * 16 * gKHb; i += blockDim.x) {
                const int c = i / (16 * gKHb), k = i - c * 16 * gKHb;
                if (k >= 16 * gHT) ZL[c * KZ + kperm(k)] = 0.f;
            }
        }
        if (dead && lane == 0) RED[24] = 1.f;
        __syncthreads();
        if constexpr (!X3) { if (RED[24] != 0.f) { alive = false; return; } }      // a wave that gave up takes the whole workgroup with it
        BSTAMP(4 + 5 * (6 - j));
        // ---- phase B ----
        f32x4 gb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            const float gave_up = RED[24];      // (requested in front of the fragments -- x3_tile's first scheduling barrier keeps it there -- and looked at behind the products)
            gb = x3_tile<4>(xB, ZX, lane);
            if (gave_up != 0.f) { alive = false; return; }      // (X3: the flag is read with the fragments, not in front of them -- an LDS round trip per stage less on the chain; a workgroup that gives up has multiplied for nothing)
        }
        else if (tile_ok) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* zb = ZL + col * KZ + 4 * (lane >> 4);
            f32x4 bf[kSMaxHT];
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) if (kb < gKHb) bf[kb] = *(const f32x4*)(zb + 16 * kb);
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) {
                if (kb < gKHb) {      // (FIX: the k-steps past row H - 1 multiply zeros and are left out -- 25 MFMAs instead of 28)
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    if (!FIX || 16 * kb + 4 < gH) acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    if (!FIX || 16 * kb + 8 < gH) acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    if (!FIX || 16 * kb + 12 < gH) acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
            }
            gb = acc0 + acc1;
            if constexpr (!FIX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (r0 + i >= gD) gb[i] = 0.f;
            }
        }
        BSTAMP(5 + 5 * (6 - j));
        // ---- phase C ----
        if constexpr (j > 1) slab_clears_done();      // before this stage's put (see slab_put; the clears were issued two phases ago)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tile_ok) {
            if (has_eig && j == 5) gb += exg;
            gbs[j - 1] = gb;
            if (j == 6) unb += gb;
            constexpr int jn = j - 1;
            f32x4 kbar = tsA(6, jn) * unb + tsBt(jn) * utb;
#pragma unroll
            for (int s = 1; s <= 5; ++s) {
                if (s > jn) kbar += tsA(s, jn) * gbs[s - 1];
            }
            if (has_sv) kbar += Wv[jn];
            kbar = dt * kbar;
            if constexpr (jn >= 1) {
                const f32x4 ks = c_ks;
#pragma unroll
                for (int i = 0; i < 4; ++i) S += ks[i] * kbar[i];
                if (has_eig && j == 6) kbar += exk;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (FIX || r0 + i < gD) ? (ACT2 ? kbar[i] * (1.f - ks[i] * ks[i]) : kbar[i]) : 0.f;
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
                    uo += first ? ld4(Bq.ubar + co, r0, gD, colok, false) : ld4(Bq.U + co, r0, gD, true, vec);
                    if (!first) ko += ld4(Bq.K1 + co, r0, gD, true, vec);
                }
                st4(Bq.U + co, r0, gD, true, vec, uo);
                st4(Bq.K1 + co, r0, gD, true, vec, ko);
            }
        }
        if (!colok) tau = 0.f;
        BSTAMP(6 + 5 * (6 - j));
        if constexpr (j > 1) phase_d(v, ex + 1u);
        pS[7 - j] = wave_sum_f(S); pT[7 - j] = wave_sum_f(tau);      // (behind the put, see START)
        BSTAMP(7 + 5 * (6 - j));
    };
    stage(std::integral_constant<int, 6>{});
    stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 1>{});
    if (!alive) return;
    BSTAMP(34);

    // ---- per-workgroup partials {S, tau, sum_j c_j tau_j (+ saveat dt-bar)}: same reduction order as the 7 launches ----
    // (no barrier in front of the GL writes below: the last readers of GL are the phase D loads of stage 2, and every wave has passed stage 1's
    //  phase A barrier since -- a wave that is done with stage 1 forms its wave sums while the slower ones finish)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float a = pS[i], b = pT[i], c = i == 0 ? pX[0] : 0.f;      // wave sums, formed behind each part's put (only START has an exdt term)
        if (lane == 0) { GL[(i * 3 + 0) * 8 + w] = a; GL[(i * 3 + 1) * 8 + w] = b; GL[(i * 3 + 2) * 8 + w] = c; }
    }
    __syncthreads();
    if (w == 0) {
        // lane 3 i + kind sums entry (i, kind) over the waves (q = 0 .. in order, as one thread did for all 21 entries: 147 dependent-chain additions
        // behind 147 LDS reads at the very end of every reversed attempt), lane 0 then combines the 21 sums in the same order as before
        float v = 0.f;
        if (lane < 21) for (int q = 0; q < gWT; ++q) v += GL[lane * 8 + q];
        auto at = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        float o0 = at(0), o1 = at(1), o2 = at(2);
#pragma unroll
        for (int i = 1; i < 7; ++i) { const float sa = at(3 * i), ta = at(3 * i + 1); o0 += sa; o1 += ta; o2 += tsC(7 - i) * ta; }
        if (lane == 0) {
            float* o = Bq.bpart + ((size_t)(n & 1) * Bq.bpart_n + wg) * 4;
            o[0] = o0; o[1] = o1; o[2] = o2; o[3] = 0.f;
        }
        BSTAMP(35);
    }
}

}  // namespace rnde
