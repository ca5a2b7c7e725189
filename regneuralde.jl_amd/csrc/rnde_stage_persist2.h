// rnde_stage_persist2.h -- the one-launch attempted step for batches that fill the chip more than once: NCT column tiles per workgroup.
//
// rnde_stage_attempt_kernel (rnde_stage_persist.h) gives every 16-column tile its own 7 workgroups.  At B = 512 that is exactly one
// workgroup per CU and the attempt is a latency chain; at B >= 1024 the chip is filled several times over and the same kernel is bound
// by its own instruction streams: per stage and tile ~3.5 k cycles of MFMA issue on the SIMDs two waves share, ~2.2 k of hand-off
// latency, two barriers, ~400 VALU operations per wave (tanh twice, stage combination, operand staging) -- and a weight slice (64 VGPRs
// per lane) loaded per tile and launch (profiles/r03_attempt_ablation.csv).  Here a workgroup owns NCT = 2 neighbouring column tiles of
// the SAME row block: the weight registers, the launch prologue (controller, 175 KB of weight loads per CU), every barrier and every
// hand-off wait serve twice the columns, and the two tiles' instruction streams are independent, so one tile's tanh / combinations
// issue under the other's MFMAs (the scheduler interleaves them inside the wave; the two waves of a SIMD do the rest).
//
// Same arithmetic in the same order per column as the one-tile kernel, the same slab protocol per tile (a tile's row blocks still meet
// on one XCD: the pair index takes the place of the tile index in the workgroup -> XCD mapping), per-tile error partials at the same
// indices: results are bit-identical to rnde_stage_attempt_kernel (tests/test_gpu_forward.py::test_two_tile_attempt_is_bit_identical).
// Headline geometry only (D = 784, H = 100, 7 waves, 7 row blocks); selected by the host when the number of column tiles is a
// multiple of NCT and at least kPersist2MinTiles.
#pragma once
#include "rnde_stage_persist.h"

namespace rnde {

constexpr int kPersist2MinTiles = 64;      // B >= 1024: below that one tile per workgroup fills more CUs

// The tiles of a workgroup ALTERNATE -- stage s of tile 0, stage s of tile 1, stage s + 1 of tile 0 ... -- so a tile's hand-off is in flight for a
// whole stage of the other tile before anybody polls it: the wait (hand-off latency and the start-up skew between a tile's row blocks, 20 % of an
// attempt at B = 4096, profiles/r03_attempt_ablation.csv) disappears behind useful work.  (Two other schedules were measured in round 3 and are kept
// under tools/experiments/stage_layouts/: lock step, 180 us against 178, and skewed -- one tile's MFMA step beside the other's element-wise step --,
// 187 us: with fp32 operands the matrix pipe and the vector ALU of a SIMD do not run in the same cycle)
// X3 = 1: the Dense-layer products on the matrix cores (rnde_x3.h) -- the same instruction sequence per tile as rnde_stage_attempt_kernel<.., 1, 1> and the x3
// one-launch solve, hence bit-identical to them.  Here the split pays twice: the matrix instructions of one tile no longer occupy the vector ALUs the
// other tile's tanh / combinations need.
template <int ACT2, int NCT, int X3 = 0>
__global__ __launch_bounds__(64 * 7) void rnde_stage_attempt_mt_kernel(const StageParams Q, const int n, const PersistSync Y, const void* x3B = nullptr, const void* x3D = nullptr) {
    const StepParams& P = Q.F;
    constexpr int gWT = 7, gHT = 7, gK2b = 7, gMT = 49, gR = 7, gD = 784, gH = 100;
    constexpr int KH = 16 * gK2b + 4, KG = 16 * gWT + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* HL = smem;                            // [NCT][16][KH]   (X3: [NCT] operand images of bf16 planes, rnde_x3.h)
    float* GL = HL + NCT * (X3 ? kX3ImageFloats : kSCB * KH);            // [NCT][16][KG]
    float* RED = GL + NCT * (X3 ? kX3ImageFloats : kSCB * KG);           // [NCT][32]; RED[24] = "a wave gave up"
    unsigned short* HX = (unsigned short*)HL;
    unsigned short* GX = (unsigned short*)GL;
    constexpr int kImgShorts = 2 * kX3ImageFloats;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < 7);
    const int C2 = Q.C / NCT;                    // (the host launches this kernel only when NCT divides the tile count)
    const int rb = (blockIdx.x >> 3) % gR, ctp = 8 * ((blockIdx.x >> 3) / gR) + (blockIdx.x & 7);
    if (ctp >= C2) return;
    const int col = lane & 15;
    int ct[NCT], wg[NCT], gcol[NCT];
    bool colok[NCT];
    size_t co[NCT];
#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) {
        ct[tt] = ctp * NCT + tt; wg[tt] = rb * Q.C + ct[tt]; gcol[tt] = ct[tt] * kSCB + col; colok[tt] = gcol[tt] < P.B;
        co[tt] = (size_t)gcol[tt] * gD;
    }
    constexpr bool vec = true;
    const bool writer = (wg[0] == 0 && tid == 0);
    const int T = rb * gWT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const RecLayout L{(long long)gD * P.Bpad, (long long)gH * P.Bpad};
    if (tid == 0) {
        const unsigned xid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID
#pragma unroll
        for (int tt = 0; tt < NCT; ++tt) Y.xcc[wg[tt]] = xid;
    }

    // ---- everything the launch reads, requested in one piece (rnde_stage_persist.h explains the order) ----
    float pre_part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 prev_raw[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (n > 0) {
        const f32x4* cp = (const f32x4*)&P.ctl[(n - 1) & 1];
        prev_raw[0] = cp[0]; prev_raw[1] = cp[1]; prev_raw[2] = cp[2];
        if (!P.esum) partials_request(P.errpart + (size_t)((n - 1) & 1) * 3 * P.nwg, lane, pre_part);      // (esum: the sums were formed behind the previous launch, rnde_epart_reduce_kernel)
    }
    const bool spec = P.tape && n > 0 && !P.forced;
    f32x4 sp_up[NCT], sp_k[NCT];
#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) {
        sp_up[tt] = (f32x4){0.f, 0.f, 0.f, 0.f}; sp_k[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (spec) {
            const float* Rs = P.arena + (long long)(n - 1) * P.rec_stride;
            sp_up[tt] = ld4(Rs + L.unew() + co[tt], r0, gD, true, vec);
            sp_k[tt] = ld4(Rs + L.k(7) + co[tt], r0, gD, true, vec);
        }
    }
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < gH) { w1t_own[i] = Q.p[(size_t)gH * gD + hr]; b1_own[i] = Q.p[(size_t)gH * (gD + 1) + hr]; }
    }
    typedef const __attribute__((address_space(1))) f32x4* gw4;
    unsigned long long aB = (unsigned long long)(Q.pwB + ((size_t)T * gK2b) * 64 + lane);
    unsigned long long aD = (unsigned long long)(Q.pwD + ((size_t)w * gMT + rb * gWT) * 64 + lane);
    unsigned long long aB4 = aB + 4 * 1024, aD4 = aD + 4 * 1024;
    asm volatile("" : "+v"(aB), "+v"(aD), "+v"(aB4), "+v"(aD4));
    f32x4 wB[X3 ? 1 : 7], wD[X3 ? 1 : 7];
    x3u4 xB[X3 ? 4 : 1][3], xD[X3 ? 4 : 1][3];
    unsigned long long xb_addr[3] = {0ull, 0ull, 0ull};
    if constexpr (X3) {
        typedef const __attribute__((address_space(1))) x3u4* gx4;
        unsigned long long bD = (unsigned long long)((const x3u4*)x3D + ((size_t)(w * gR + rb) * 4 * 3) * 64 + lane);
        unsigned long long bB = (unsigned long long)((const x3u4*)x3B + ((size_t)T * 4 * 3) * 64 + lane);
        unsigned long long bD1 = bD + 4 * 1024, bD2 = bD + 8 * 1024, bB1 = bB + 4 * 1024, bB2 = bB + 8 * 1024;
        asm volatile("" : "+v"(bD), "+v"(bB), "+v"(bD1), "+v"(bD2), "+v"(bB1), "+v"(bB2));
#pragma unroll
        for (int f = 0; f < 12; ++f) xD[f / 3][f % 3] = f < 4 ? ((gx4)bD)[(size_t)f * 64] : (f < 8 ? ((gx4)bD1)[(size_t)(f - 4) * 64] : ((gx4)bD2)[(size_t)(f - 8) * 64]);
        xb_addr[0] = bB; xb_addr[1] = bB1; xb_addr[2] = bB2;      // (xB: first multiplied in stage 1's phase B -- requested behind the controller and the state the step starts from, see there)
        // k-values 112 .. 135 of every (image, plane, column) row are written by nobody: zeroed (only those: no barrier before START's x3_store4)
        for (int i = tid; i < 2 * NCT * 3 * 16 * 12; i += 64 * 7) ((unsigned*)HL)[(i / 12) * (kX3K / 2) + 56 + i % 12] = 0u;
    } else {
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) wD[kb] = kb < 4 ? ((gw4)aD)[(size_t)kb * 64] : ((gw4)aD4)[(size_t)(kb - 4) * 64];
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (kb == 6) {
            typedef const __attribute__((address_space(1))) f32x2* gw2;
            const f32x2 lo = *(gw2)(aB4 + 2 * 1024);
            wB[kb] = (f32x4){lo.x, lo.y, 0.f, 0.f};
        } else wB[kb] = kb < 4 ? ((gw4)aB)[(size_t)kb * 64] : ((gw4)aB4)[(size_t)(kb - 4) * 64];
    }
    }

    // ---- controller (identical to SM_START) ----
    asm volatile("" : "+v"(prev_raw[0]), "+v"(prev_raw[1]), "+v"(prev_raw[2]));
    StepState prev_state;
    __builtin_memcpy(&prev_state, prev_raw, sizeof(StepState));
    prev_state.live = __builtin_amdgcn_readfirstlane(prev_state.live); prev_state.done = __builtin_amdgcn_readfirstlane(prev_state.done);
    const StepState S = advance_state_t<true>(P, n, lane, writer, &P.ctl[n & 1], pre_part, prev_state, (P.esum && n > 0) ? P.esum + 4 * ((n - 1) & 1) : nullptr);
    if (P.nsave > 0) {
        const int lo = (n == 0) ? 0 : P.ctl[(n - 1) & 1].next_save, hi = S.next_save;
        if (hi > lo) {
#pragma unroll
            for (int tt = 0; tt < NCT; ++tt) {
                if (n == 0) {
                    st_tile(P.sv_out + (size_t)gcol[tt] * P.nsave * gD, r0, gD, colok[tt], vec, ld_tile(P.x + co[tt], r0, gD, colok[tt], P.xvec != 0));
                } else {
                    const StepState pv = P.ctl[(n - 1) & 1];
                    const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                    const float* Rp = P.arena + (long long)S.live * P.rec_stride;
                    dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, co[tt], gcol[tt], r0, colok[tt], vec);
                }
            }
        }
    }
    if (S.done) return;
    const float t = S.t, dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
    const int live = S.live;
    const int rec = P.tape ? n + P.rec_shift : (live == 0 ? 1 : 0);
    float* R = P.arena + (long long)rec * P.rec_stride;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upvec = P.xvec != 0;
    if (live >= 0) { const float* Rl = P.arena + (long long)live * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upvec = vec; }
    f32x4 c_up[NCT], c_un[NCT], c_k[NCT][6];
#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) {
        c_un[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; ++j) c_k[tt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (spec && live == n - 1) { c_up[tt] = sp_up[tt]; c_k[tt][0] = sp_k[tt]; }
        else { c_up[tt] = ld4(upsrc + co[tt], r0, gD, live >= 0 ? true : colok[tt], upvec); c_k[tt][0] = ld4(k1p + co[tt], r0, gD, true, vec); }
    }
    if constexpr (X3) {
        // the weight fragments of phase B at the END of the prologue's request queue (a wave's loads return in order; round 6, rnde_bstage_persist.h: the same move
        // took 0.9 us off a reversed attempt): START's own phase D waits for xD, the controller's inputs and the state only
        typedef const __attribute__((address_space(1))) x3u4* gx4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 12; ++f) xB[f / 3][f % 3] = ((gx4)xb_addr[f >> 2])[(size_t)(f & 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }

    // loop-invariant addressing of this lane's four rows of its hidden tile (phase A) and row tile (phase D); offsets inside ONE tile's image
    const int hr0 = 16 * w + 4 * (lane >> 4);
    const int own_hl0 = col * KH + kperm(hr0), own_gl0 = col * KG + kperm(hr0);
    int own_kind[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int hr = hr0 + i; own_kind[i] = hr < gH ? 0 : (hr == gH ? 1 : (hr == gH + 1 ? 2 : 3)); }
    float own_c1[4], own_c0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { own_c1[i] = own_kind[i] == 1 ? 1.f : 0.f; own_c0[i] = own_kind[i] == 2 ? 1.f : 0.f; }
    const bool own_hstore = rb == w && own_kind[3] == 0;      // (row block rb tapes hidden tile rb, see rnde_stage_solve.h)
    if (tid == 0) RED[24] = 0.f;

    float part0[NCT], part1[NCT], part2[NCT];
#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) part0[tt] = part1[tt] = part2[tt] = 0.f;
    bool alive = true;


    // phase D for all tiles: this row block's layer-1 partials of the stage inputs v[tt] -> slab, exchange number `ex`
    auto phase_d = [&](const f32x4 (&v)[NCT], unsigned ex, auto t0c, auto t1c) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
        if constexpr (X3) {
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) x3_store4(GX + tt * kImgShorts, col, hr0, v[tt]);
            __syncthreads();
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                const size_t tile0x = (((size_t)slab_buf(ex) * Q.C + ct[tt]) * gR + rb) * gHT;
                slab_put(Y.tslab, tile0x + w, lane, x3_tile<4>(xD, GX + tt * kImgShorts, lane));
            }
            return;
        }
#pragma unroll
        for (int tt = T0; tt < T1; ++tt) {
            float* gl = GL + tt * kSCB * KG + own_gl0;
#pragma unroll
            for (int i = 0; i < 4; ++i) gl[4 * i] = v[tt][i];
        }
        __syncthreads();
        f32x4 acc0[NCT], acc1[NCT];
#pragma unroll
        for (int tt = T0; tt < T1; ++tt) { acc0[tt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[tt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) {
            f32x4 bg[NCT];
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) bg[tt] = *(const f32x4*)(GL + tt * kSCB * KG + col * KG + 4 * (lane >> 4) + 16 * kb);
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                acc0[tt] = mfma16(wD[kb][0], bg[tt][0], acc0[tt]);
                acc1[tt] = mfma16(wD[kb][1], bg[tt][1], acc1[tt]);
            }
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                acc0[tt] = mfma16(wD[kb][2], bg[tt][2], acc0[tt]);
                acc1[tt] = mfma16(wD[kb][3], bg[tt][3], acc1[tt]);
            }
        }
#pragma unroll
        for (int tt = T0; tt < T1; ++tt) {
            const size_t tile0 = (((size_t)slab_buf(ex) * Q.C + ct[tt]) * gR + rb) * gHT;
            slab_put(Y.tslab, tile0 + w, lane, acc0[tt] + acc1[tt]);
        }
    };

    // ---- SM_START's phase C / D ----
    {
        f32x4 v[NCT];
#pragma unroll
        for (int tt = 0; tt < NCT; ++tt) {
            v[tt] = fma4(dt, tsA(1, 0) * c_k[tt][0], c_up[tt]);
            if (P.tape) st4(R + L.g(2) + co[tt], r0, gD, true, vec, v[tt]);
            if (P.nsave > 0) { st4(R + L.upc() + co[tt], r0, gD, true, vec, c_up[tt]); st4(R + L.k1c() + co[tt], r0, gD, true, vec, c_k[tt][0]); }
        }
        phase_d(v, 1u, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (NCT > 1) phase_d(v, 1u, std::integral_constant<int, 1>{}, std::integral_constant<int, NCT>{});
    }

    auto stage = [&](auto sc, auto t0c, auto t1c) {
        constexpr int s = decltype(sc)::value, T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
        if (!alive) return;
        const float ts = fmaf(tsC(s), dt, t);
        float* hdst = R + L.h(s + 1);
        float* kdst = R + L.k(s + 1);
        const int buf = slab_buf((unsigned)s);
        // ---- phase A: poll this wave's hidden tile of the R row blocks, for every column tile ----
        bool dead = false;
        f32x4 zs[NCT];
#pragma unroll
        for (int tt = T0; tt < T1; ++tt) {
            zs[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!dead) dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct[tt], w, lane, zs[tt]);
        }
#pragma unroll
        for (int tt = T0; tt < T1; ++tt) {
            const size_t tprev0 = (((size_t)slab_buf((unsigned)(s + 2)) * Q.C + ct[tt]) * gR + rb) * gHT;
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = fmaf(w1t_own[i], ts, zs[tt][i]) + b1_own[i];
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            f32x4 hv = {t01.x, t01.y, t23.x, t23.y};
            if (w == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = own_kind[i] == 0 ? hv[i] : fmaf(own_c1[i], ts, own_c0[i]);
            }
            if (own_hstore) *(f32x4*)(hdst + (size_t)gcol[tt] * gH + hr0) = hv;
            if constexpr (X3) x3_store4(HX + tt * kImgShorts, col, hr0, hv);
            else {
                float* hl = HL + tt * kSCB * KH + own_hl0;
#pragma unroll
                for (int i = 0; i < 4; ++i) hl[4 * i] = hv[i];
            }
        }
        if (dead && lane == 0) RED[24] = 1.f;
        __syncthreads();
        if constexpr (!X3) { if (RED[24] != 0.f) { alive = false; return; } }
        // ---- phase B ----
        f32x4 kv[NCT];
        if constexpr (X3) {
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                const float gave_up = RED[24];      // (requested in front of the fragments -- x3_tile's first scheduling barrier keeps it there -- and looked at behind the products)
                kv[tt] = x3_tile<4>(xB, HX + tt * kImgShorts, lane);
                if (gave_up != 0.f) { alive = false; return; }      // (X3: the flag is read with the fragments, not in front of them -- an LDS round trip per stage less on the chain; a workgroup that gives up has multiplied for nothing)
                if (ACT2) {
                    const f32x2 a01 = tanh_fast2((f32x2){kv[tt][0], kv[tt][1]}), a23 = tanh_fast2((f32x2){kv[tt][2], kv[tt][3]});
                    kv[tt] = (f32x4){a01.x, a01.y, a23.x, a23.y};
                }
            }
        } else {
            f32x4 acc0[NCT], acc1[NCT];
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) { acc0[tt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[tt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kb = 0; kb < 7; ++kb) {
                f32x4 bf[NCT];
#pragma unroll
                for (int tt = T0; tt < T1; ++tt) bf[tt] = *(const f32x4*)(HL + tt * kSCB * KH + col * KH + 4 * (lane >> 4) + 16 * kb);
#pragma unroll
                for (int tt = T0; tt < T1; ++tt) {
                    acc0[tt] = mfma16(wB[kb][0], bf[tt][0], acc0[tt]);
                    if (16 * kb + 4 < gH + 2) acc1[tt] = mfma16(wB[kb][1], bf[tt][1], acc1[tt]);
                }
#pragma unroll
                for (int tt = T0; tt < T1; ++tt) {
                    if (16 * kb + 8 < gH + 2) acc0[tt] = mfma16(wB[kb][2], bf[tt][2], acc0[tt]);
                    if (16 * kb + 12 < gH + 2) acc1[tt] = mfma16(wB[kb][3], bf[tt][3], acc1[tt]);
                }
            }
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                kv[tt] = acc0[tt] + acc1[tt];
                if (ACT2) {
                    const f32x2 a01 = tanh_fast2((f32x2){kv[tt][0], kv[tt][1]}), a23 = tanh_fast2((f32x2){kv[tt][2], kv[tt][3]});
                    kv[tt] = (f32x4){a01.x, a01.y, a23.x, a23.y};
                }
            }
        }
        // ---- phase C ----
        if constexpr (s < 6) {
            slab_clears_done();
            f32x4 v[NCT];
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                st4(kdst + co[tt], r0, gD, true, vec, kv[tt]);
                f32x4 acc = tsA(s + 1, 0) * c_k[tt][0];
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(tsA(s + 1, j), c_k[tt][j], acc);
                acc = fma4(tsA(s + 1, s), kv[tt], acc);
                v[tt] = fma4(dt, acc, c_up[tt]);
                if (s == 5) { st4(R + L.unew() + co[tt], r0, gD, true, vec, v[tt]); c_un[tt] = v[tt]; }
                else if (P.tape) st4(R + L.g(s + 2) + co[tt], r0, gD, true, vec, v[tt]);
                c_k[tt][s] = kv[tt];
            }
            phase_d(v, (unsigned)(s + 1), t0c, t1c);
        } else {
#pragma unroll
            for (int tt = T0; tt < T1; ++tt) {
                st4(kdst + co[tt], r0, gD, true, vec, kv[tt]);
                const f32x4 up = c_up[tt], un = c_un[tt];
                f32x4 acc = tsBt(0) * c_k[tt][0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = fma4(tsBt(j), c_k[tt][j], acc);
                acc = fma4(tsBt(6), kv[tt], acc);
                if (colok[tt]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ut = dt * acc[i];
                        const float sk = P.abstol + fmaxf(fabsf(up[i]), fabsf(un[i])) * P.reltol;
                        const float r = ut / sk;
                        part0[tt] = add_square_unfused(part0[tt], r);
                    }
                    if (P.reg_kind >= 2) {
                        f32x4 g6 = tsA(5, 0) * c_k[tt][0];
#pragma unroll
                        for (int j = 1; j < 5; ++j) g6 = fma4(tsA(5, j), c_k[tt][j], g6);
                        g6 = fma4(dt, g6, up);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float d1 = kv[tt][i] - c_k[tt][5][i], d2 = un[i] - g6[i];
                            part1[tt] = add_square_unfused(part1[tt], d1); part2[tt] = add_square_unfused(part2[tt], d2);
                        }
                    }
                }
            }
        }
    };
    auto both = [&](auto sc) {
        stage(sc, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (NCT > 1) stage(sc, std::integral_constant<int, 1>{}, std::integral_constant<int, NCT>{});
    };
    both(std::integral_constant<int, 1>{});
    both(std::integral_constant<int, 2>{});
    both(std::integral_constant<int, 3>{});
    both(std::integral_constant<int, 4>{});
    both(std::integral_constant<int, 5>{});
    both(std::integral_constant<int, 6>{});
    if (!alive) return;

#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) {
        const float a = wave_sum_f(part0[tt]), b = wave_sum_f(part1[tt]), c = wave_sum_f(part2[tt]);
        if (lane == 0) { RED[32 * (tt + 1) + w] = a; RED[32 * (tt + 1) + 8 + w] = b; RED[32 * (tt + 1) + 16 + w] = c; }
    }
    __syncthreads();
    if (tid < NCT) {
        const float* rd = RED + 32 * (tid + 1);
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int i = 0; i < gWT; ++i) { sa += rd[i]; sb += rd[8 + i]; sc += rd[16 + i]; }
        float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
        const int g = rb * Q.C + ctp * NCT + tid;
        ep[g] = sa; ep[P.nwg + g] = sb; ep[2 * P.nwg + g] = sc;
    }
}

}  // namespace rnde
