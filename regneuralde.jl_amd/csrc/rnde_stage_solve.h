// rnde_stage_solve.h -- the stage engine's WHOLE adaptive solve as ONE launch (headline geometry: D = 784, H = 100, 7 x 7 row tiles).
//
// rnde_stage_attempt_kernel (rnde_stage_persist.h) already runs the seven stages of an attempted step in one launch; what it still pays
// per attempt is the kernel boundary the error norm needs: launch ramp, the controller's cold loads (state, 224 partials), 64 VGPRs of
// weights per lane re-read from the Infinity Cache, the state arrays (uprev, k1) re-read from the tape -- ~4 us of a 27 us attempt at
// B = 512, 39 times per forward solve (reference call site: `solve(prob, Tsit5(); ...)`, src/models/neural_ode.jl:131-137).
// Here the attempt loop itself is in the kernel:
//   * the weight slice, w1t / b1, uprev and k1 (= unew and k7 of the last accepted attempt) never leave registers;
//   * the error norm meets through memory: after stage 7 every workgroup publishes its partial sums as 8-byte {value, tag} granules
//     (ONE agent-scope store each: the data is its own validity, `cdna_hip_programming.md` Guideline 16, form R2), wave 0 of every
//     workgroup sweeps all of them with agent-scope loads and forms the sums in the order `sum_partials` forms them -- so the solve is
//     bit-identical to the one-launch-per-attempt path (tests/test_gpu_solve.py);
//   * the PI controller (advance_state_t, rnde_fwd.h) runs between the meeting and the next attempt's first stage, as it does in the
//     prologue of the attempt kernel; StepMeta / StepState records are written exactly as before (the reverse pass reads them).
// The workgroups of a column tile still hand the layer-1 partials to each other through their XCD's L2 (slab_put / slab_poll_sum);
// the meeting crosses XCDs, which is why it uses agent-scope (sc1) stores and loads on granules every workgroup writes once per solve
// (entry index = attempt number, tag = epoch * 8192 + attempt + 1: nothing is ever reused inside a launch, nothing needs clearing).
// All workgroups must be resident at once (<= 256, one per CU); every spin is bounded, a time-out raises the abort word the attempt
// kernels use and the host redoes the solve launch by launch.
#pragma once
#ifndef RNDE_SOLVE_DEFER_TAPE
#define RNDE_SOLVE_DEFER_TAPE 1
#endif
#ifndef RNDE_SOLVE_LOCAL_LAYOUT
#define RNDE_SOLVE_LOCAL_LAYOUT 1
#endif
#include "rnde_stage_persist.h"
#include "rnde_solve_sync.h"
#include "rnde_x3.h"

namespace rnde {

typedef __attribute__((address_space(1))) unsigned long long solve_gu64;

// Cross-workgroup sums of the norm partials of attempt `seq`.  Called by wave 0 of every workgroup; `mine` valid in lane 0.  The sums come
// out as sum_partials forms them: lane l adds entries l, l + 64, l + 128, l + 192 in that order in double, then the wave reduction.
// nval = 1 (error norm) or 3 (+ the two norms of the stiffness estimate).  false: timed out / aborted.
__device__ __forceinline__ bool solve_meet(const SolveSync& Z, const PersistSync& Y, int seq, int nwg, int wg, int nval, const float (&mine)[3],
                                           double (&out)[3], int lane) {
    const unsigned tag = Z.epoch * 8192u + (unsigned)seq + 1u;
    solve_gu64* base = (solve_gu64*)Z.xch + (size_t)seq * 3 * 256;
    if (lane == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            if (v < nval) {
                float m = mine[v];
                if (m != m) m = __uint_as_float(0x7FC00000u);
                __hip_atomic_store(base + (size_t)v * 256 + wg, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(m), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    out[0] = out[1] = out[2] = 0.0;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        if (v >= nval) break;
        // (one set of polling loads at a time: keeping a second set in flight -- measured in round 4 -- makes an attempt 0.3 us SLOWER; the extra
        //  reads of the same lines on the memory side delay the stores they are waiting for; a back-off between polls, s_sleep 4 / 8 / 16, does not help
        //  either: 23.37 / 23.46 / 23.60 us against 23.34)
        unsigned long long e[4] = {0, 0, 0, 0};
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ok[q] = lane + 64 * q >= nwg;
        int spins = 0;
        while (true) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!ok[q]) {
                    e[q] = __hip_atomic_load(base + (size_t)v * 256 + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok[q] = (unsigned)(e[q] >> 32) == tag;
                }
            }
            if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
            if (++spins > Y.max_spins || ((spins & 255) == 0 && __hip_atomic_load(Y.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) __hip_atomic_store(Y.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        double s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane + 64 * q < nwg) s += (double)__uint_as_float((unsigned)(e[q] & 0xFFFFFFFFull));
        out[v] = wave_sum_d(s);
    }
    return true;
}

// the controller state through LDS: written by one lane, read back by every wave as scalars (twelve words)
static_assert(sizeof(StepState) == 48, "StepState is twelve 4-byte words");
__device__ __forceinline__ void solve_state_put(int* ss, const StepState& S) {
    const int* src = (const int*)&S;
#pragma unroll
    for (int i = 0; i < 12; ++i) ss[i] = src[i];
}
__device__ __forceinline__ StepState solve_state_get(const int* ss) {
    StepState S;
    int* dst = (int*)&S;
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[i] = __builtin_amdgcn_readfirstlane(ss[i]);
    return S;
}

// X3 = 1: the two Dense-layer products of every stage on the matrix cores (rnde_x3.h: exact three-way bf16 split of both operands, six
// v_mfma_f32_16x16x32_bf16 per 32 k-values) instead of the fp32-input MFMA, which this part executes on its vector ALUs.  Everything else -- hand-off,
// meeting, controller, tape layout, the state in registers -- is the same code.  The products are rounded differently (more accurately: rnde_x3.h), so
// an X3 solve is NOT bit-identical to the fp32-MFMA kernels; its parity is stated against the fp64 restatement (tests/test_gpu_x3.py).
template <int ACT2, int X3>
__global__ __launch_bounds__(64 * 7) void rnde_stage_solve_kernel(const StageParams Q, const PersistSync Y, const SolveSync Z) {
    const StepParams& P = Q.F;
    constexpr int gWT = 7, gHT = 7, gK2b = 7, gR = 7, gD = 784, gH = 100;
    constexpr int KH = 16 * gK2b + 4, KG = 16 * gWT + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* HL = smem;
    float* GL = HL + (X3 ? kX3ImageFloats : kSCB * KH);
    float* RED = GL + (X3 ? kX3ImageFloats : kSCB * KG);         // [32]; RED[24] = "a wave of this workgroup gave up"
    unsigned short* HX = (unsigned short*)HL;      // X3: the operand images [plane][column][kX3K] of bf16 in place of the fp32 images
    unsigned short* GX = (unsigned short*)GL;
    double* SUMS = (double*)(RED + 32);  // [4]: the three cross-workgroup sums of the meeting, [3] != 0: the meeting failed
    float* QP = (float*)(SUMS + 4);      // [2] (by attempt parity: a fast wave writes the next one while a slow wave still reads this one): powf(qold, beta2) of the state the running attempt started from (evaluated off the critical path)
    int* SS = (int*)(QP + 2);            // [12] the controller state before the next attempt, as wave 0 derived it behind the meeting (see solve_state_put / _get)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < 7);
    const int rb = (blockIdx.x >> 3) % gR, ct = 8 * ((blockIdx.x >> 3) / gR) + (blockIdx.x & 7);      // a tile's row blocks share blockIdx % 8 (one XCD)
    if (ct >= Q.C) return;
    const int wg = rb * Q.C + ct;
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    constexpr bool vec = true;
    const bool writer = (wg == 0 && tid == 0);
    const int T = rb * gWT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
#if RNDE_SOLVE_LOCAL_LAYOUT
    const RecLayout LL{(long long)gD * P.Bpad, (long long)gH * P.Bpad};
#else
    const RecLayout L{(long long)gD * P.Bpad, (long long)gH * P.Bpad};
#endif
    if (tid == 0) Y.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID
    const size_t co = (size_t)gcol * gD;

    // ---- once per solve: the state the first attempt starts from, this block's weight slice, the bias / time column of this wave's hidden tile ----
    f32x4 c_up = ld4(P.x + co, r0, gD, colok, P.xvec != 0), c_k[7];
    c_k[0] = ld4(P.f0 + co, r0, gD, true, vec);
#pragma unroll
    for (int j = 1; j < 7; ++j) c_k[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < gH) { w1t_own[i] = Q.p[(size_t)gH * gD + hr]; b1_own[i] = Q.p[(size_t)gH * (gD + 1) + hr]; }
    }
    f32x4 wB[X3 ? 1 : 7], wD[X3 ? 1 : 7];
    x3u4 xB[X3 ? 4 : 1][3], xD[X3 ? 4 : 1][3];
    if constexpr (X3) {
        const x3u4* pB = (const x3u4*)Z.x3B + ((size_t)T * 4 * 3) * 64 + lane;
        const x3u4* pD = (const x3u4*)Z.x3D + ((size_t)(w * gR + rb) * 4 * 3) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { xD[s][pl] = pD[(size_t)(s * 3 + pl) * 64]; xB[s][pl] = pB[(size_t)(s * 3 + pl) * 64]; }
        }
        // the operand images start as zeros: k-values nobody writes (102 .. 127 of the hidden layer's, 112 .. 127 of the row block's) multiply zero weights
        for (int i = tid; i < 2 * kX3ImageFloats; i += 64 * 7) HL[i] = 0.f;
    } else {
        const f32x4* pB = Q.pwB + ((size_t)T * gK2b) * 64 + lane;
        const f32x4* pD = Q.pwD + ((size_t)w * 49 + rb * gWT) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) wD[kb] = pD[(size_t)kb * 64];
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) wB[kb] = pB[(size_t)kb * 64];
        wB[6][2] = 0.f; wB[6][3] = 0.f;      // (k-steps 104.. multiply zeros and are left out, as in the attempt kernel)
    }

    // loop-invariant addressing of this lane's four rows of its own hidden tile (phase A) and row tile (phase D)
    int own_kind[4];
    float own_c1[4], own_c0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        own_kind[i] = hr < gH ? 0 : (hr == gH ? 1 : (hr == gH + 1 ? 2 : 3));
        own_c1[i] = own_kind[i] == 1 ? 1.f : 0.f; own_c0[i] = own_kind[i] == 2 ? 1.f : 0.f;
    }
    const int own_hl0 = col * KH + kperm(16 * w + 4 * (lane >> 4));
    const int own_gl0 = col * KG + kperm(16 * w + 4 * (lane >> 4));
    const size_t own_hd0 = (size_t)gcol * gH + 16 * w + 4 * (lane >> 4);
    // (the hidden activations are computed identically by all seven row blocks of a tile: row block rb tapes hidden tile rb -- one store per
    //  stage and workgroup instead of seven in row block 0, whose workgroups would otherwise reach every meeting last)
    const bool own_hstore = rb == w && own_kind[3] == 0;
    if (tid == 0) { RED[24] = 0.f; SUMS[0] = SUMS[1] = SUMS[2] = SUMS[3] = 0.0; }
    __syncthreads();

    // phase D: this row block's layer-1 partial of the stage input v -> slab, exchange number `ex`
    auto phase_d = [&](const f32x4& v, unsigned ex) {
        if constexpr (X3) {
            x3_store4(GX, col, 16 * w + 4 * (lane >> 4), v);
            __syncthreads();
            const size_t tile0x = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
            slab_put(Y.tslab, tile0x + w, lane, x3_tile<4>(xD, GX, lane));
            return;
        }
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

#ifdef RNDE_DIAG      // cycle stamps of workgroup 0, eight per attempt (tools/diag_solve.py)
#define ZSTAMP(k) do { if (P.dbg_out && wg == 0 && tid == 0 && n < 120) ((unsigned long long*)P.dbg_out)[n * 8 + (k)] = clock64(); } while (0)
// every workgroup's arrival at the meeting of attempt 10 and its way out of it (s_memrealtime: one clock for all XCDs, 100 MHz)
#define ZARRIVE(k) do { if (P.dbg_out && tid == 0 && n == 10) ((unsigned long long*)P.dbg_out)[1024 + wg * 2 + (k)] = wall_clock64(); } while (0)
#else
#define ZSTAMP(k) do { } while (0)
#define ZARRIVE(k) do { } while (0)
#endif
    // ---- controller: ONE wave per workgroup derives the state before an attempt (wave 0: it holds the meeting's sums in registers and is alone on its
    // SIMD while wave 4 waits at the barrier), the others pick the twelve words up from LDS.  Round 4 had every wave derive it redundantly from the
    // sums in LDS: the same function on the same arguments -- the same bits -- at seven times the vector instructions (the controller's float arithmetic,
    // two pow and the double-precision norms, runs on the vector ALU: this part has no scalar float unit), two waves per SIMD taking turns.
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    StepState S{};
    if (w == 0) {
        const StepState nop{};
        S = advance_state_t<true>(P, 0, lane, writer, &P.ctl[0], none, nop, nullptr, nullptr);
        if (lane == 0) solve_state_put(SS, S);
    }
    __syncthreads();
    S = solve_state_get(SS);
    f32x4 c_un = {0.f, 0.f, 0.f, 0.f};
    // the tape stores of a stage (k_{s+1}, g_{s+2} / u_new) are issued BEHIND the next stage's poll: a wave's polling load cannot complete before every
    // older store of the wave is acknowledged (vmcnt counts both), so a store in front of a poll puts its acknowledgement on the hand-off's critical path;
    // behind the poll it has a whole stage to come back.  The values are in registers anyway (c_k[s]; the stage input is carried in `pend`).  Matrix mode 1
    // only: there it takes 0.3 us off an attempt in-step; the fp32-input-MFMA form's stages got 3 % LONGER with it (41.8 k against 40.5 k cycles, stamps).
    constexpr bool DEFER = RNDE_SOLVE_DEFER_TAPE && X3;
    f32x4 pend = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0;; ++n) {
        ZSTAMP(0);
        // the controller of the NEXT attempt divides by qold^beta2, and qold is known now: wave 3 -- alone on its SIMD -- evaluates the power while
        // the stages run, the others pick it up after the meeting (the barriers in between order the LDS word); same function, same argument, same bits
        if (w == 3 && lane == 0) QP[(n + 1) & 1] = powf(S.qold, P.beta2);
        ZSTAMP(1);
        if (S.done || n >= Z.n_limit) { if (writer) *P.ctl_final = S; return; }
        const float t = S.t, dt = (P.t1 - S.t < S.dtp) ? (P.t1 - S.t) : S.dtp;
        const int rec = P.tape ? n : (S.live == 0 ? 1 : 0);
        float* R = P.arena + (long long)rec * P.rec_stride;
#if RNDE_SOLVE_LOCAL_LAYOUT
        // the record's array offsets are formed from (A, HB) where they are used, every attempt: left to itself the compiler keeps the ~19 loop-invariant 64-bit
        // offsets in scalar registers across the attempt loop and spills them (75-81 SGPR spills, 143 v_readlane per attempt); the empty asm hides the invariance
        long long A_n = LL.A, HB_n = LL.HB;
        asm volatile("" : "+s"(A_n), "+s"(HB_n));
        const RecLayout L{A_n, HB_n};
#endif

        // ---- the attempt kernel's START: g2, exchange 1 ----
        {
            const f32x4 v = fma4(dt, tsA(1, 0) * c_k[0], c_up);
            // the clears of the previous attempt's last stage are acknowledged before this attempt's first put -- they were issued before the meeting, so
            // nothing waits here; BEHIND the tape store below the same wait would sit on that store's acknowledgement (vmcnt counts stores too)
            slab_clears_done();
            if constexpr (DEFER) pend = v;
            else if (P.tape) st4(R + L.g(2) + co, r0, gD, true, vec, v);
            phase_d(v, 1u);
        }
        ZSTAMP(2);

        float part0 = 0.f, part1 = 0.f, part2 = 0.f;
        bool alive = true;
        auto stage = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (!alive) return;
            const float ts = fmaf(tsC(s), dt, t);
            float* hdst = R + L.h(s + 1);
            float* kdst = R + L.k(s + 1);
            const int buf = slab_buf((unsigned)s);
            // ---- phase A: poll this wave's hidden tile of the 7 row blocks (the polling load is the data load) ----
            f32x4 zs = {0.f, 0.f, 0.f, 0.f};
            const bool dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, w, lane, zs);
            const size_t tprev0 = (((size_t)slab_buf((unsigned)(s + 2)) * Q.C + ct) * gR + rb) * gHT;
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            if constexpr (DEFER) {
                if constexpr (s >= 2) st4(R + L.k(s) + co, r0, gD, true, vec, c_k[s - 1]);
                if constexpr (s == 6) st4(R + L.unew() + co, r0, gD, true, vec, pend);
                else if (P.tape) st4(R + L.g(s + 1) + co, r0, gD, true, vec, pend);
            }
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = fmaf(w1t_own[i], ts, zs[i]) + b1_own[i];
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            f32x4 hv = {t01.x, t01.y, t23.x, t23.y};
            if (w == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = own_kind[i] == 0 ? hv[i] : fmaf(own_c1[i], ts, own_c0[i]);
            }
            if (own_hstore) *(f32x4*)(hdst + own_hd0) = hv;
            if constexpr (X3) x3_store4(HX, col, 16 * w + 4 * (lane >> 4), hv);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) HL[own_hl0 + 4 * i] = hv[i];
            }
            if (dead && lane == 0) RED[24] = 1.f;
            __syncthreads();
            if constexpr (!X3) { if (RED[24] != 0.f) { alive = false; return; } }
            // ---- phase B ----
            f32x4 kv;
            if constexpr (X3) {
                const float gave_up = RED[24];      // (requested in front of the fragments -- x3_tile's first scheduling barrier keeps it there -- and looked at behind the products)
                kv = x3_tile<4>(xB, HX, lane);
                if (gave_up != 0.f) { alive = false; return; }      // (X3: the flag is read with the fragments, not in front of them -- an LDS round trip per stage less on the chain; a workgroup that gives up has multiplied for nothing)
                if (ACT2) {
                    const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                    kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
                }
            } else {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* hb = HL + col * KH + 4 * (lane >> 4);
                f32x4 bf[7];
#pragma unroll
                for (int kb = 0; kb < 7; ++kb) bf[kb] = *(const f32x4*)(hb + 16 * kb);
#pragma unroll
                for (int kb = 0; kb < 7; ++kb) {
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    if (16 * kb + 4 < gH + 2) acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    if (16 * kb + 8 < gH + 2) acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    if (16 * kb + 12 < gH + 2) acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
                kv = acc0 + acc1;
                if (ACT2) {
                    const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                    kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
                }
            }
            // ---- phase C ----
            if constexpr (s < 6) {
                slab_clears_done();      // (issued two phases ago: nothing to wait for in practice) before this stage's put, see slab_put
                if constexpr (!DEFER) st4(kdst + co, r0, gD, true, vec, kv);
                f32x4 acc = tsA(s + 1, 0) * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(tsA(s + 1, j), c_k[j], acc);
                acc = fma4(tsA(s + 1, s), kv, acc);
                const f32x4 v = fma4(dt, acc, c_up);
                if constexpr (DEFER) { pend = v; if (s == 5) c_un = v; }
                else {
                    if (s == 5) { st4(R + L.unew() + co, r0, gD, true, vec, v); c_un = v; }
                    else if (P.tape) st4(R + L.g(s + 2) + co, r0, gD, true, vec, v);
                }
                c_k[s] = kv;
                phase_d(v, (unsigned)(s + 1));
            } else {
                st4(kdst + co, r0, gD, true, vec, kv);
                c_k[6] = kv;      // k7: k1 of the next attempt if this one is accepted
                const f32x4 up = c_up, un = c_un;
                f32x4 acc = tsBt(0) * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = fma4(tsBt(j), c_k[j], acc);
                acc = fma4(tsBt(6), kv, acc);
                if (colok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ut = dt * acc[i];
                        const float sk = P.abstol + fmaxf(fabsf(up[i]), fabsf(un[i])) * P.reltol;
                        const float r = ut / sk;
                        part0 = add_square_unfused(part0, r);
                    }
                    if (P.reg_kind >= 2) {
                        f32x4 g6 = tsA(5, 0) * c_k[0];
#pragma unroll
                        for (int j = 1; j < 5; ++j) g6 = fma4(tsA(5, j), c_k[j], g6);
                        g6 = fma4(dt, g6, up);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float d1 = kv[i] - c_k[5][i], d2 = un[i] - g6[i];
                            part1 = add_square_unfused(part1, d1); part2 = add_square_unfused(part2, d2);
                        }
                    }
                }
            }
        };
        stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{});
        stage(std::integral_constant<int, 4>{});
        stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{});
        if (!alive) return;
        ZSTAMP(3);

        // ---- the meeting: this workgroup's partials (same reduction order as the attempt kernel), everybody's sums ----
        const bool three = P.reg_kind >= 2;      // (the two norms of the stiffness estimate travel only when it is asked for)
        part0 = wave_sum_f(part0);
        if (three) { part1 = wave_sum_f(part1); part2 = wave_sum_f(part2); }
        if (lane == 0) { RED[w] = part0; if (three) { RED[8 + w] = part1; RED[16 + w] = part2; } }
        __syncthreads();
        ZSTAMP(4); ZARRIVE(0);
        if (w == 0) {
            float mine[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < gWT; ++i) { mine[0] += RED[i]; if (three) { mine[1] += RED[8 + i]; mine[2] += RED[16 + i]; } }
            double o[3];
            const bool ok = solve_meet(Z, Y, n, P.nwg, wg, P.reg_kind >= 2 ? 3 : 1, mine, o, lane);
            ZSTAMP(5); ZARRIVE(1);
            // the state before attempt n + 1, from the sums just formed (qold^beta2 of the state this attempt started from: wave 3 left it in QP while
            // the stages ran; the barrier in front of the meeting ordered it)
            const float qp = QP[(n + 1) & 1];
            const StepState nxt = advance_state_t<true>(P, n + 1, lane, writer, &P.ctl[(n + 1) & 1], none, S, o, &qp);
            if (lane == 0) { solve_state_put(SS, nxt); if (!ok) SUMS[3] = 1.0; }
        }
        __syncthreads();
        ZSTAMP(6);
        if (SUMS[3] != 0.0) return;      // the meeting timed out: abort word raised, the host redoes the solve launch by launch
        {
            const StepState nxt = solve_state_get(SS);
            if (nxt.n_acc != S.n_acc) { c_up = c_un; c_k[0] = c_k[6]; }      // accepted: the next attempt starts from (unew, k7) -- already here
            S = nxt;
        }
    }
}

}  // namespace rnde
