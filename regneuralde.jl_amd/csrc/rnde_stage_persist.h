// rnde_stage_persist.h -- the stage engine's attempted step as ONE launch.
//
// rnde_stage_kernel (rnde_stage.h) crosses a kernel boundary after every Runge-Kutta stage only to hand the split-K
// partials of layer 1 (the "slab") from the R row blocks of a column tile to each other.  Each boundary costs the launch
// ramp plus a chain of cold, dependent loads (controller state, weights, state arrays): ~9 us per stage, of which ~3 us is
// arithmetic.  Here the 7 stages of an attempt run in one kernel:
//   * weights (this block's 1/R slice, 64 VGPRs per lane) and the attempt's own k_1..k_6 / uprev rows stay in registers;
//   * the slab hand-off happens inside the kernel between the R workgroups of a column tile.  They have the same
//     blockIdx % 8, hence sit on the same XCD (round-robin dispatch; verified at run time from HW_REG_XCC_ID) and share its
//     L2: producer = slab stores, workgroup-scope release (s_waitcnt vmcnt(0): the write-through stores are in L2),
//     barrier, relaxed agent-scope flag store; consumer = one wave polls the R flags with L1-bypassing loads, agent-scope
//     acquire (buffer_inv sc1), barrier, plain loads.  Measured 3.1 us per hand-off (tools/micro/cluster_sync.hip).
//     No L2 write-back is involved, which is what would make an agent-scope release slow on a multi-XCD part;
//   * the kernel boundary that remains (one per attempt) is the one the algorithm needs: the global error norm.
// Arithmetic, association order and tape layout are exactly those of rnde_stage_kernel, so results are bit-identical
// (tests/test_gpu_forward.py::test_persistent_attempt_is_bit_identical).
// Every spin is bounded: on time-out (workgroups not co-resident, or a placement that is not what was assumed) the
// kernel raises `abort_flag`, all workgroups leave, and the host falls back to the multi-launch kernels for good.
#pragma once
#include "rnde_stage.h"

#include <type_traits>

namespace rnde {

struct PersistSync {
    unsigned* flags;        // [C][8] sequence numbers, monotonic across launches
    unsigned* abort_flag;   // [0] abort, [1] placement error (cluster spans XCDs)
    unsigned* xcc;          // [grid] XCC id of each workgroup (written every launch)
    unsigned seq_base;
};

constexpr int kPersistMaxSpins = 200000;

__device__ __forceinline__ void persist_signal(const PersistSync& Y, int ct, int rb, unsigned seq, int tid) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's slab stores have reached L2
    __syncthreads();
    if (tid == 0) __hip_atomic_store(Y.flags + ct * 8 + rb, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false when the launch must be abandoned (uniform over the workgroup)
__device__ __forceinline__ bool persist_wait(const PersistSync& Y, int ct, int R, unsigned seq, int w, int lane, float* RED) {
    if (w == 0) {
        unsigned v = seq;
        int spins = 0;
        bool dead = false;
        while (true) {
            if (lane < R) v = __hip_atomic_load(Y.flags + ct * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = (lane >= R) || ((int)(v - seq) >= 0);
            if (__all(ok)) break;
            if (++spins > kPersistMaxSpins || __hip_atomic_load(Y.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(Y.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's L1 lines: the slab addresses are reused every other stage
        if (lane == 0) RED[31] = dead ? 1.f : 0.f;
    }
    __syncthreads();
    return RED[31] == 0.f;
}

template <int ACT2>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_stage_attempt_kernel(const StageParams Q, const int n, const PersistSync Y) {
    const StepParams& P = Q.F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KH = 16 * Q.K2b + 4, KG = 16 * Q.WT + 4;
    float* HL = smem;
    float* GL = HL + kSCB * KH;
    float* RED = GL + kSCB * KG;         // [32]; RED[31] = abandon flag of persist_wait
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (row block, column tile): the R row blocks of a column tile must sit on ONE XCD (they talk through its
    // L2).  Dispatch is round-robin, XCD = blockIdx % 8, so a tile's members are given block indices that agree mod 8;
    // the grid is 8 * R * ceil(C / 8) and the surplus workgroups (ct >= C) leave at once.
    const int rb = (blockIdx.x >> 3) % Q.R, ct = 8 * ((blockIdx.x >> 3) / Q.R) + (blockIdx.x & 7);
    if (ct >= Q.C) return;
    const int wg = rb * Q.C + ct;        // logical workgroup id (index of the per-workgroup partials, as in rnde_stage_kernel)
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (P.D & 3) == 0;
    const bool writer = (wg == 0 && tid == 0);
    const int T = rb * Q.WT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const bool tile_ok = T < Q.MT;
    const RecLayout L{(long long)P.D * P.Bpad, (long long)P.H * P.Bpad};
    if (tid == 0) Y.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID

    // ---- this block's weight slice and the layer-1 bias / time column of this wave's hidden tile: loaded once ----
    f32x4 wB[kSMaxHT], wD[kSMaxW];
#pragma unroll
    for (int kb = 0; kb < kSMaxHT; ++kb)
        if (kb < Q.K2b && tile_ok) wB[kb] = Q.pwB[((size_t)T * Q.K2b + kb) * 64 + lane];
#pragma unroll
    for (int kb = 0; kb < kSMaxW; ++kb)
        if (kb < Q.WT && w < Q.HT && rb * Q.WT + kb < Q.MT) wD[kb] = Q.pwD[((size_t)w * Q.MT + rb * Q.WT + kb) * 64 + lane];
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < P.H) { w1t_own[i] = Q.p[(size_t)P.H * P.D + hr]; b1_own[i] = Q.p[(size_t)P.H * (P.D + 1) + hr]; }
    }
    const float* W1t = Q.p + (size_t)P.H * P.D;
    const float* b1 = Q.p + (size_t)P.H * (P.D + 1);

    // ---- controller (identical to SM_START) ----
    const StepState S = advance_state(P, n, lane, writer, &P.ctl[n & 1]);
    if (P.nsave > 0) {
        const int lo = (n == 0) ? 0 : P.ctl[(n - 1) & 1].next_save, hi = S.next_save;
        if (hi > lo && tile_ok) {
            if (n == 0) {
                st_tile(P.sv_out + (size_t)gcol * P.nsave * P.D, r0, P.D, colok, vec, ld_tile(P.x + (size_t)gcol * P.D, r0, P.D, colok, P.xvec != 0));
            } else {
                const StepState pv = P.ctl[(n - 1) & 1];
                const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                const float* Rp = P.arena + (long long)S.live * P.rec_stride;
                dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, (size_t)gcol * P.D, gcol, r0, colok, vec);
            }
        }
    }
    if (S.done) return;
    const float t = S.t, dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
    const int live = S.live;
    const int rec = P.tape ? n : (live == 0 ? 1 : 0);
    float* R = P.arena + (long long)rec * P.rec_stride;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (live >= 0) { const float* Rl = P.arena + (long long)live * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }
    const size_t co = (size_t)gcol * P.D;
    f32x4 c_up = {0.f, 0.f, 0.f, 0.f}, c_un = {0.f, 0.f, 0.f, 0.f}, c_k[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) c_k[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tile_ok) { c_up = ld4(upsrc + co, r0, P.D, upok, upvec); c_k[0] = ld4(k1p + co, r0, P.D, true, vec); }

    // phase D: this row block's layer-1 partial of the stage input v -> slab[par], then publish exchange number `ex`
    auto phase_d = [&](const f32x4& v, int par, unsigned ex) {
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[col * KG + kperm(16 * w + 4 * (lane >> 4) + i)] = (tile_ok && r0 + i < P.D) ? v[i] : 0.f;
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
                    const f32x4 a = Q.pwD[((size_t)ht * Q.MT + rb * Q.WT + kb) * 64 + lane];
                    acc0 = mfma16(a[0], bg[kb][0], acc0);
                    acc1 = mfma16(a[1], bg[kb][1], acc1);
                    acc0 = mfma16(a[2], bg[kb][2], acc0);
                    acc1 = mfma16(a[3], bg[kb][3], acc1);
                }
            }
            sl[(size_t)ht * 64 + lane] = acc0 + acc1;
        }
        persist_signal(Y, ct, rb, Y.seq_base + ex, tid);
    };

    // ---- SM_START's phase C / D ----
    {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tile_ok) {
            v = fma4(dt, kFwdShift[0][0] * c_k[0], c_up);
            if (P.tape) st4(R + L.g(2) + co, r0, P.D, true, vec, v);
            if (P.tape || P.nsave > 0) { st4(R + L.upc() + co, r0, P.D, true, vec, c_up); st4(R + L.k1c() + co, r0, P.D, true, vec, c_k[0]); }
        }
        phase_d(v, 1, 1u);
    }

    float part0 = 0.f, part1 = 0.f, part2 = 0.f;
    bool alive = true;
    // one stage: s = 1..5 -> SM_STAGE, s = 6 -> SM_LAST (zero-based stage index as in rnde_stage_kernel)
    auto stage = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (!alive) return;
        if (!persist_wait(Y, ct, Q.R, Y.seq_base + (unsigned)s, w, lane, RED)) { alive = false; return; }
        const float ts = fmaf(kTsC[s], dt, t);
        float* hdst = R + L.h(s + 1);
        float* kdst = R + L.k(s + 1);
        const int par = s & 1;
        const f32x4* sl = (const f32x4*)Q.slab + (((size_t)par * Q.C + ct) * Q.R) * Q.HT * 64;
        // ---- phase A ----
        f32x4 zs = {0.f, 0.f, 0.f, 0.f};
        if (w < Q.HT) {
            f32x4 zr[kSMaxW];
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zr[r] = sl[((size_t)r * Q.HT + w) * 64 + lane];
#pragma unroll
            for (int r = 0; r < kSMaxW; ++r) if (r < Q.R) zs += zr[r];
            for (int r = kSMaxW; r < Q.R; ++r) zs += sl[((size_t)r * Q.HT + w) * 64 + lane];
        }
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
                float v = 0.f;
                if (hr < P.H) {
                    v = tanh_fast(fmaf((ht == w) ? w1t_own[i] : W1t[hr], ts, z[i]) + ((ht == w) ? b1_own[i] : b1[hr]));
                    if (rb == 0) hdst[(size_t)gcol * P.H + hr] = v;
                } else if (hr == P.H) v = ts;
                else if (hr == P.H + 1) v = 1.f;
                if (hr < 16 * Q.K2b) HL[col * KH + kperm(hr)] = v;
            }
        }
        if (Q.K2b > Q.HT) {
            for (int i = tid; i < kSCB * 16 * Q.K2b; i += blockDim.x) {
                const int c = i / (16 * Q.K2b), k = i - c * 16 * Q.K2b;
                if (k >= 16 * Q.HT) HL[c * KH + kperm(k)] = (k == P.H) ? ts : (k == P.H + 1 ? 1.f : 0.f);
            }
        }
        __syncthreads();
        // ---- phase B ----
        f32x4 kv = {0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
            for (int i = 0; i < 4; ++i) kv[i] = (r0 + i < P.D) ? act_apply_fast(ACT2, kv[i]) : 0.f;
        }
        // ---- phase C ----
        if constexpr (s < 6) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tile_ok) {
                st4(kdst + co, r0, P.D, true, vec, kv);
                f32x4 acc = tsA_rt(s + 1, 0) * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(tsA_rt(s + 1, j), c_k[j], acc);
                acc = fma4(tsA_rt(s + 1, s), kv, acc);
                v = fma4(dt, acc, c_up);
                if (s == 5) { st4(R + L.unew() + co, r0, P.D, true, vec, v); c_un = v; }
                else if (P.tape) st4(R + L.g(s + 2) + co, r0, P.D, true, vec, v);
                c_k[s] = kv;
            }
            phase_d(v, (s + 1) & 1, (unsigned)(s + 1));
        } else {
            if (tile_ok) {
                st4(kdst + co, r0, P.D, true, vec, kv);
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
                        part0 += r * r;
                    }
                    if (P.reg_kind >= 2) {
                        f32x4 g6 = tsA_rt(5, 0) * c_k[0];
#pragma unroll
                        for (int j = 1; j < 5; ++j) g6 = fma4(tsA_rt(5, j), c_k[j], g6);
                        g6 = fma4(dt, g6, up);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (r0 + i < P.D) {
                                const float d1 = kv[i] - c_k[5][i], d2 = un[i] - g6[i];
                                part1 += d1 * d1; part2 += d2 * d2;
                            }
                        }
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

    part0 = wave_sum_f(part0); part1 = wave_sum_f(part1); part2 = wave_sum_f(part2);
    if (lane == 0) { RED[w] = part0; RED[8 + w] = part1; RED[16 + w] = part2; }
    __syncthreads();
    if (tid == 0) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int i = 0; i < Q.WT; ++i) { sa += RED[i]; sb += RED[8 + i]; sc += RED[16 + i]; }
        float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
        ep[wg] = sa; ep[P.nwg + wg] = sb; ep[2 * P.nwg + wg] = sc;
    }
}

}  // namespace rnde
