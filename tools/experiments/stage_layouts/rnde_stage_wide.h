// rnde_stage_wide.h -- the attempted step of the MNIST form for batches that fill the chip with COLUMN TILES alone (B >= 2048 per GPU):
// one workgroup owns a 16-column tile for ALL 784 rows.
//
// The stage engine (rnde_stage_persist.h) splits a column tile over 7 row-block workgroups because at the reference's batch (512 = 32
// tiles) that is the only way to occupy the chip; the price is the hand-off of the layer-1 partials through L2 six times per attempt, the
// hidden activations computed seven times over, seven prologues per tile, and -- with two waves per SIMD and no co-execution of fp32 MFMAs
// with vector instructions -- an instruction stream in which the matrix instructions are 45 % of the issue cycles.  From 128
// column tiles on there are enough tiles to give every CU its own, and none of that is needed:
//   * no exchange between workgroups at all: no slabs, no polls, no placement assumption (works on a partitioned GPU), no co-residency;
//   * tanh of the hidden layer once per tile; one controller prologue per tile;
//   * the weights (0.68 MB per f evaluation and workgroup) stream from L2 through a register ring while the MFMAs run: wave w owns hidden
//     tile w (layer 1: 7 row blocks x 7 k-blocks = 196 MFMAs) and row tiles w, w + 7, ..., w + 42 (layer 2: 7 x 26 MFMAs); the stage input
//     (784 x 16, 50 KB) and the hidden activations (7 KB) are the LDS-resident B operands;
//   * the attempt's state (uprev, k_1 .. k_s) does not fit the registers at 49 row tiles per workgroup: a row tile's operands of the
//     stage combination are re-read from the tape (the thread's own earlier stores) while that tile's MFMAs run.
// The arithmetic is that of rnde_stage_attempt_kernel in the same order -- layer 1 as 7 row-block partials of two interleaved accumulators,
// added in order r = 0..6; layer 2 over [h; t; 1]; explicit fma chains in the combinations; per-row-block error partials at the same
// indices -- so states, step log, saved values and tape are bit-identical (tests/test_gpu_forward.py::test_wide_attempt_is_bit_identical),
// and the reverse pass (rnde_bstage_attempt_kernel) runs on its tape unchanged.  Headline geometry only (D = 784, H = 100).
//
// STATUS (round 3): correct and bit-identical, NOT selected automatically (RNDE_WIDE=1 opts in).  At B = 4096 it takes 221 us per attempted
// step against 178 us for the two-tile row-block kernels.  Ablations (tools/ab_wide.py, compile-time switches below; results wrong by
// construction): without the state re-reads and tape stores 120-128 us, with the weight blocks loaded once 126-131 us, with neither
// 98-104 us -- the instruction-issue floor of this layout (378 MFMAs + ~1,800 vector instructions per wave and stage), which is where
// the design pays off.  Either memory stream alone is hidden; together they are not: a wave's vector-memory operations complete in
// order, a row tile's 26 MFMAs (0.4-0.8 us) do not cover the ~1 us its combination operands take to come back from the Infinity Cache
// (the XCD's share of the tape, 14 MB, is past its 4 MB L2), and with two 28-register weight blocks in flight there is no room to request
// them a tile earlier (256 VGPRs + 512 B of scratch already).  And behind the latency there is a bandwidth wall: per attempt a workgroup
// moves 4.1 MB of weights + 1.7 MB of combination operands + 0.85 MB of tape stores = 6.7 MB through its CU's one path to L2 (64 B/clk
// = 134 GB/s at best; ~70 GB/s was what a CU sustained when round 1's column-owner kernels streamed weights) -- 50-95 us of pure transfer
// against a 100 us compute floor, so even a perfect ring (weights by DMA into the 98 KB of LDS that GL / HL leave free, operands requested a
// tile ahead) lands near 150 us at best.  The weight-stationary row-block kernels move 1/7 of a tile's weights ONCE per attempt per CU;
// that is why they win, and why this layout stays an experiment.
#pragma once
#include "rnde_stage_persist.h"

namespace rnde {

constexpr int kWideMinTiles = 128;      // B >= 2048: below that the two-tile stage kernels fill more CUs (not used while the kernel is opt-in)
constexpr int kWideKG = 16 * 49 + 4;    // LDS column stride of the stage-input image (same bank pattern as the stage engine's 16 * 7 + 4)
constexpr int kWideKH = 16 * 7 + 4;

template <int ACT2>
__global__ __launch_bounds__(64 * 7) void rnde_stage_wide_kernel(const StageParams Q, const int n, const PersistSync Y) {
    const StepParams& P = Q.F;
    constexpr int gD = 784, gH = 100, KG = kWideKG, KH = kWideKH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* GL = smem;                    // [16][KG]  stage input, all 784 rows, k permuted inside every 16-block
    float* HL = GL + kSCB * KG;          // [16][KH]  hidden activations (+ t, 1)
    float* RED = HL + kSCB * KH;         // [3][7 row blocks][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < 7);
    const int ct = blockIdx.x;
    const int col = lane & 15, g4 = lane >> 4, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool writer = (ct == 0 && tid == 0);
    const RecLayout L{(long long)gD * P.Bpad, (long long)gH * P.Bpad};
    const size_t co = (size_t)gcol * gD;
    if (tid < 7) Y.xcc[tid * Q.C + ct] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // (one XCD per tile by construction: the host's placement check passes)

    // ---- controller (identical to SM_START): state and error partials of attempt n - 1 requested first ----
    float pre_part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 prev_raw[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (n > 0) {
        const f32x4* cp = (const f32x4*)&P.ctl[(n - 1) & 1];
        prev_raw[0] = cp[0]; prev_raw[1] = cp[1]; prev_raw[2] = cp[2];
        partials_request(P.errpart + (size_t)((n - 1) & 1) * 3 * P.nwg, lane, pre_part);
    }
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * g4 + i;
        if (hr < gH) { w1t_own[i] = Q.p[(size_t)gH * gD + hr]; b1_own[i] = Q.p[(size_t)gH * (gD + 1) + hr]; }
    }
    // weight blocks of this wave, in the order an evaluation consumes them: 0..6 = layer 1, row block b (hidden tile w, k-blocks 7 b .. 7 b + 6);
    // 7..13 = layer 2, row tile 7 (b - 7) + w (k-blocks 0 .. 6).  A block = 7 x 16 bytes per lane = 7 KiB of consecutive memory per wave.
    const f32x4* wb1 = Q.pwD + ((size_t)w * 49) * 64 + lane;
    const f32x4* wb2 = Q.pwB + ((size_t)w * 7) * 64 + lane;
    auto load_blk = [&](int b, f32x4 (&a)[7]) {
#ifdef RNDE_WIDE_ABL_NOWEIGHTS
        if (b > 1) return;
#endif
        const f32x4* src = b < 7 ? wb1 + (size_t)(7 * b) * 64 : wb2 + (size_t)(49 * (b - 7)) * 64;
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) a[kb] = src[(size_t)kb * 64];
    };
    f32x4 wa[7], wc[7];                  // the ring: block being multiplied / block in flight

    asm volatile("" : "+v"(prev_raw[0]), "+v"(prev_raw[1]), "+v"(prev_raw[2]));
    StepState prev_state;
    __builtin_memcpy(&prev_state, prev_raw, sizeof(StepState));
    prev_state.live = __builtin_amdgcn_readfirstlane(prev_state.live); prev_state.done = __builtin_amdgcn_readfirstlane(prev_state.done);
    const StepState S = advance_state_t<true>(P, n, lane, writer, &P.ctl[n & 1], pre_part, prev_state);
    if (P.nsave > 0) {
        const int lo = (n == 0) ? 0 : P.ctl[(n - 1) & 1].next_save, hi = S.next_save;
        if (hi > lo) {
            for (int i = 0; i < 7; ++i) {
                const int r0 = 16 * (7 * i + w) + 4 * g4;
                if (n == 0) st_tile(P.sv_out + (size_t)gcol * P.nsave * gD, r0, gD, colok, true, ld_tile(P.x + co, r0, gD, colok, P.xvec != 0));
                else {
                    const StepState pv = P.ctl[(n - 1) & 1];
                    const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                    const float* Rp = P.arena + (long long)S.live * P.rec_stride;
                    dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, co, gcol, r0, colok, true);
                }
            }
        }
    }
    if (S.done) return;
    const float t = S.t, dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
    const int live = S.live;
    const int rec = P.tape ? n + P.rec_shift : (live == 0 ? 1 : 0);
    float* R = P.arena + (long long)rec * P.rec_stride;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (live >= 0) { const float* Rl = P.arena + (long long)live * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = true; }

    int own_kind[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int hr = 16 * w + 4 * g4 + i; own_kind[i] = hr < gH ? 0 : (hr == gH ? 1 : (hr == gH + 1 ? 2 : 3)); }
    float own_c1[4], own_c0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { own_c1[i] = own_kind[i] == 1 ? 1.f : 0.f; own_c0[i] = own_kind[i] == 2 ? 1.f : 0.f; }
    const bool own_hstore = own_kind[3] == 0;
    float* hl_own = HL + col * KH + kperm(16 * w + 4 * g4);
    const float* hb = HL + col * KH + 4 * g4;
    const float* gb = GL + col * KG + 4 * g4;

    load_blk(0, wa);                     // (behind the controller and the dense output: in front of them the 28 registers spill)
    // ---- START: g_2 = uprev + dt a_21 k_1 for this wave's seven row tiles -> GL (and the tape) ----
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
        const int r0 = 16 * (7 * i + w) + 4 * g4;
        const f32x4 up = ld_tile(upsrc + co, r0, gD, upok, upvec), k1 = *(const f32x4*)(k1p + co + r0);
        const f32x4 v = fma4(dt, tsA(1, 0) * k1, up);
        if (P.tape) *(f32x4*)(R + L.g(2) + co + r0) = v;
        if (P.nsave > 0) { *(f32x4*)(R + L.upc() + co + r0) = up; *(f32x4*)(R + L.k1c() + co + r0) = k1; }
        float* gl = GL + col * KG + 16 * (7 * i + w) + g4;       // kperm: row 16 T + 4 g4 + i sits at 16 T + 4 i + g4
#pragma unroll
        for (int q = 0; q < 4; ++q) gl[4 * q] = v[q];
    }
    __syncthreads();

#pragma unroll 1
    for (int s = 1; s <= 6; ++s) {       // zero-based stage index as in rnde_stage_kernel: k_{s+1} = f(g_{s+1}, t + c_s dt)
        const float ts = fmaf(kTsC[s], dt, t);
        // ---- layer 1: hidden tile w as seven row-block partials (two interleaved accumulators each), added in order ----
        f32x4 zs = {0.f, 0.f, 0.f, 0.f};
        auto l1_block = [&](const f32x4 (&a)[7], int b) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 7; ++kb) {
                const f32x4 bg = *(const f32x4*)(gb + 16 * (7 * b + kb));
                acc0 = mfma16(a[kb][0], bg[0], acc0);
                acc1 = mfma16(a[kb][1], bg[1], acc1);
                acc0 = mfma16(a[kb][2], bg[2], acc0);
                acc1 = mfma16(a[kb][3], bg[3], acc1);
            }
            const f32x4 f = acc0 + acc1;
            // (slab_put's clamp of the one bit pattern that means "empty" is the identity on everything arithmetic produces here)
            zs[0] += f[0]; zs[1] += f[1]; zs[2] += f[2]; zs[3] += f[3];
#ifndef RNDE_WIDE_NOSB
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        load_blk(1, wc); l1_block(wa, 0);
        load_blk(2, wa); l1_block(wc, 1);
        load_blk(3, wc); l1_block(wa, 2);
        load_blk(4, wa); l1_block(wc, 3);
        load_blk(5, wc); l1_block(wa, 4);
        load_blk(6, wa); l1_block(wc, 5);
        load_blk(7, wc); l1_block(wa, 6);
        {
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = fmaf(w1t_own[i], ts, zs[i]) + b1_own[i];
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            f32x4 hv = {t01.x, t01.y, t23.x, t23.y};
            if (w == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = own_kind[i] == 0 ? hv[i] : fmaf(own_c1[i], ts, own_c0[i]);
            }
            if (own_hstore) *(f32x4*)(R + L.h(s + 1) + (size_t)gcol * gH + 16 * w + 4 * g4) = hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) hl_own[4 * i] = hv[i];
        }
        __syncthreads();
        // ---- layer 2 + stage combination, row tile by row tile; the next stage's input goes straight into GL ----
        f32x4 bf[7];
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) bf[kb] = *(const f32x4*)(hb + 16 * kb);
        auto l2_tile = [&](const f32x4 (&a)[7], int i) {
            const int r0 = 16 * (7 * i + w) + 4 * g4;
            // this tile's operands of the combination: requested now, used after the MFMAs
#ifdef RNDE_WIDE_ABL_NOSTATE
            f32x4 c_up = bf[0], c_k[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) c_k[j] = bf[j];
            f32x4 c_un = bf[6];
#else
            f32x4 c_up = ld_tile(upsrc + co, r0, gD, upok, upvec), c_k[6];
            c_k[0] = *(const f32x4*)(k1p + co + r0);
#pragma unroll
            for (int j = 1; j < 6; ++j) c_k[j] = j < s ? *(const f32x4*)(R + L.k(j + 1) + co + r0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 c_un = s == 6 ? *(const f32x4*)(R + L.unew() + co + r0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 7; ++kb) {
                acc0 = mfma16(a[kb][0], bf[kb][0], acc0);
                if (16 * kb + 4 < gH + 2) acc1 = mfma16(a[kb][1], bf[kb][1], acc1);
                if (16 * kb + 8 < gH + 2) acc0 = mfma16(a[kb][2], bf[kb][2], acc0);
                if (16 * kb + 12 < gH + 2) acc1 = mfma16(a[kb][3], bf[kb][3], acc1);
            }
            f32x4 kv = acc0 + acc1;
            if (ACT2) {
                const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
            }
#ifndef RNDE_WIDE_ABL_NOSTATE
            *(f32x4*)(R + L.k(s + 1) + co + r0) = kv;
#endif
            if (s < 6) {
                f32x4 acc = kTsA[s + 1][0] * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(kTsA[s + 1][j], c_k[j], acc);
                acc = fma4(kTsA[s + 1][s], kv, acc);
                const f32x4 v = fma4(dt, acc, c_up);
#ifndef RNDE_WIDE_ABL_NOSTATE
                if (s == 5) *(f32x4*)(R + L.unew() + co + r0) = v;
                else if (P.tape) *(f32x4*)(R + L.g(s + 2) + co + r0) = v;
#endif
                float* gl = GL + col * KG + 16 * (7 * i + w) + g4;
#pragma unroll
                for (int q = 0; q < 4; ++q) gl[4 * q] = v[q];
            } else {
                f32x4 acc = kTsBt[0] * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = fma4(kTsBt[j], c_k[j], acc);
                acc = fma4(kTsBt[6], kv, acc);
                float p0 = 0.f, p1 = 0.f, p2 = 0.f;
                if (colok) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float ut = dt * acc[q];
                        const float sk = P.abstol + fmaxf(fabsf(c_up[q]), fabsf(c_un[q])) * P.reltol;
                        const float r = ut / sk;
                        p0 = add_square_unfused(p0, r);
                    }
                    if (P.reg_kind >= 2) {
                        f32x4 g6 = kTsA[5][0] * c_k[0];
#pragma unroll
                        for (int j = 1; j < 5; ++j) g6 = fma4(kTsA[5][j], c_k[j], g6);
                        g6 = fma4(dt, g6, c_up);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float d1 = kv[q] - c_k[5][q], d2 = c_un[q] - g6[q];
                            p1 = add_square_unfused(p1, d1); p2 = add_square_unfused(p2, d2);
                        }
                    }
                }
                // per row block as the stage engine forms them: errpart[rb * C + ct] = sum over the block's 7 row tiles (= the 7 waves) of the wave sums
                const float sa = wave_sum_f(p0), sb = wave_sum_f(p1), sc = wave_sum_f(p2);
                if (lane == 0) { RED[(0 * 7 + i) * 8 + w] = sa; RED[(1 * 7 + i) * 8 + w] = sb; RED[(2 * 7 + i) * 8 + w] = sc; }
            }
#ifndef RNDE_WIDE_NOSB
            __builtin_amdgcn_sched_barrier(0);      // (a tile's loads and arithmetic stay inside the tile: hoisting the next tiles' operand loads spills)
#endif
        };
        // (wc holds block 7; the ring runs on through the seven row tiles and into the next stage's first layer-1 block)
        load_blk(8, wa);  l2_tile(wc, 0);
        load_blk(9, wc);  l2_tile(wa, 1);
        load_blk(10, wa); l2_tile(wc, 2);
        load_blk(11, wc); l2_tile(wa, 3);
        load_blk(12, wa); l2_tile(wc, 4);
        load_blk(13, wc); l2_tile(wa, 5);
        load_blk(0, wa);  l2_tile(wc, 6);
        __syncthreads();
    }

    // ---- the error-norm partials (RED was filled by the last stage; its closing barrier has been passed) ----
    if (tid < 7) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int q = 0; q < 7; ++q) { sa += RED[(0 * 7 + tid) * 8 + q]; sb += RED[(1 * 7 + tid) * 8 + q]; sc += RED[(2 * 7 + tid) * 8 + q]; }
        float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
        const int g = tid * Q.C + ct;
        ep[g] = sa; ep[P.nwg + g] = sb; ep[2 * P.nwg + g] = sc;
    }
}

}  // namespace rnde
