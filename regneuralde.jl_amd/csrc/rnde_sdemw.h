// rnde_sdemw.h -- the SDE engine's whole-solve kernel with FOUR waves per 16-column tile, for the reference's own NSDE shape
// (experiments/mnist_nsde.jl:73-74: D <= 32, drift Dense(D, H <= 64) -> Dense(H, D), diffusion Dense(D, D)).
//
// rnde_sde_solve_kernel (rnde_sde.h) gives a tile to ONE wave: 80 MFMAs, 16 tanh registers and the stage combinations of 8-register
// arrays in a single instruction stream, ~9.4 k cycles per stage, 8 CUs busy at B = 512.  Here a tile is a workgroup of four waves:
//   * weights are register stationary per wave for the whole solve (drift layer 1: one of its four output tiles per wave; then waves
//     0, 1 hold the two output tiles of drift layer 2 and waves 2, 3 the two of the diffusion, which run side by side);
//   * activations cross the waves through LDS as [feature][16 columns] (an MFMA B operand is one conflict-free ds_read_b32, its D
//     registers go back to their feature rows): three barriers per stage;
//   * everything element-wise (stage combinations, noise operations, tape, saveat) is 2 registers per array and thread,
//     element e = tid + 256 r <-> (feature e >> 4, column e & 15) -- which IS the fragment order of the one-wave engine, so slot
//     pool, tape, reverse kernel, parameter-gradient kernel and the packed weight tables are shared unchanged.
// Controller, RSwM3 bookkeeping (sde_decide), exchange protocol and replay are those of rnde_sde.h.  Same arithmetic up to the
// association order of the dot products.
#pragma once
#include "rnde_sde.h"

namespace rnde {

constexpr int kSmwThreads = 256;
constexpr int kSmwMaxTiles = 512;    // workgroups of one solve launch (two per CU): every one of them must be resident, they meet once per attempt
constexpr int kSmwLdsFloats = 3072;     // X0, X1, HD, KO, GO

__global__ __launch_bounds__(kSmwThreads) void rnde_sde_solve_mw_kernel(const SdeParams Q) {
    constexpr int NKD = 2;      // registers per array and thread: element e = tid + 256 r <-> (feature e >> 4, column e & 15)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // xch_local: the grid is 8 x nwg and only the workgroups with blockIdx % 8 == 0 work -- under round-robin dispatch all on one XCD (each
    // records its XCC id; the host verifies and falls back to the placement-independent exchange if the assumption ever fails)
    if (Q.xch_local && (blockIdx.x & 7)) return;
    const int wg = Q.xch_local ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (Q.xch_local && tid == 0) Q.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID
    float* X0 = smem;                 // [32][16] drift input of the stage
    float* X1 = smem + 512;           // [32][16] diffusion input
    float* HD = smem + 1024;          // [64][16] hidden layer of the drift
    float* KO = smem + 2048;          // [32][16] drift output
    float* GO = smem + 2560;          // [32][16] diffusion output
    float* scratch = smem + 3072;
    float* RED = scratch;                                   // [4][4]: error-norm partials, initial-step partials, the two norms of the stiffness estimate
    SdeDecision* DEC = (SdeDecision*)(scratch + 16);       // 16-byte aligned
    SdeStacks* STK = (SdeStacks*)(scratch + 16 + 32);
    SdeOp* OPS = (SdeOp*)(scratch + 16 + 32 + 8);
    const int cap = 2 * Q.max_attempts + 8;
    float* S1L = (float*)(OPS + kSdeMaxOps);
    int* S1s = (int*)(S1L + cap);
    float* S2L = (float*)(S1s + cap);
    int* S2s = (int*)(S2L + cap);
    int* FREEL = (int*)(S2s + cap);
    // this wave's weight fragments, register stationary for the whole solve (the one-wave engine's packed tables: A operand of output
    // tile mo, k-step k of layer l at [(foff[l] + mo nks[l] + k)][lane]; output register i of tile mo is feature 16 mo + 4 i + (lane >> 4))
    const int lg = lane >> 4, lc = lane & 15;
    float a1[8], b1[4], a2[16], b2[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) a1[k] = Q.frags_f[((size_t)Q.Gf.foff[0] + wave * 8 + k) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) b1[i] = Q.frags_f[((size_t)Q.Gf.nfrag_f + Q.Gf.boff[0] + 4 * wave + i) * 64 + lane];
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a2[k] = Q.frags_f[((size_t)Q.Gf.foff[1] + wave * 16 + k) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) b2[i] = Q.frags_f[((size_t)Q.Gf.nfrag_f + Q.Gf.boff[1] + 4 * wave + i) * 64 + lane];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) a2[k] = k < 8 ? Q.frags_g[((size_t)Q.Gg.foff[0] + (wave - 2) * 8 + (k < 8 ? k : 0)) * 64 + lane] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) b2[i] = Q.frags_g[((size_t)Q.Gg.nfrag_f + Q.Gg.boff[0] + 4 * (wave - 2) + i) * 64 + lane];
    }
    const bool th1 = Q.Gf.act[0] != 0, th2 = wave < 2 ? (Q.Gf.act[1] != 0) : (Q.Gg.act[0] != 0);
    if (tid == 0) { SdeStacks s{}; *STK = s; }
    __syncthreads();

    // drift(X0) and diffusion(X1) of the 16 columns, inputs already in LDS and a barrier behind them.  Phase 1: wave w = hidden tile w of
    // the drift (8 MFMAs, bias, tanh).  Phase 2: waves 0, 1 = the two output tiles of the drift (16 MFMAs); waves 2, 3 = the two output
    // tiles of the diffusion (8 MFMAs).  Outputs come back as this thread's two elements of each.
    auto eval = [&](float (&ko)[NKD], float (&go)[NKD]) {
        {
            const float* xb = X0 + lane;            // B operand of k-step ks: feature 4 ks + lg, column lc = X[(4 ks + lg) 16 + lc] = X[64 ks + lane]
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = xb[64 * k];
            f32x4 acc0 = {b1[0], b1[1], b1[2], b1[3]}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; k += 2) { acc0 = mfma16(a1[k], bv[k], acc0); acc1 = mfma16(a1[k + 1], bv[k + 1], acc1); }
            f32x4 h = acc0 + acc1;
            if (th1) { const f32x2 t01 = tanh_fast2((f32x2){h[0], h[1]}), t23 = tanh_fast2((f32x2){h[2], h[3]}); h = (f32x4){t01.x, t01.y, t23.x, t23.y}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) HD[(16 * wave + 4 * i + lg) * 16 + lc] = h[i];
        }
        __syncthreads();
        {
            const float* xb = (wave < 2 ? HD : X1) + lane;
            float bv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) bv[k] = (k < 8 || wave < 2) ? xb[64 * (k < 8 || wave < 2 ? k : 0)] : 0.f;
            f32x4 acc0 = {b2[0], b2[1], b2[2], b2[3]}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; k += 2) { acc0 = mfma16(a2[k], bv[k], acc0); acc1 = mfma16(a2[k + 1], bv[k + 1], acc1); }
            if (wave < 2) {
#pragma unroll
                for (int k = 8; k < 16; k += 2) { acc0 = mfma16(a2[k], bv[k], acc0); acc1 = mfma16(a2[k + 1], bv[k + 1], acc1); }
            }
            f32x4 o = acc0 + acc1;
            if (th2) { const f32x2 t01 = tanh_fast2((f32x2){o[0], o[1]}), t23 = tanh_fast2((f32x2){o[2], o[3]}); o = (f32x4){t01.x, t01.y, t23.x, t23.y}; }
            float* dst = wave < 2 ? KO : GO;
            const int mo = wave & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(16 * mo + 4 * i + lg) * 16 + lc] = o[i];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NKD; ++r) { ko[r] = KO[tid + 256 * r]; go[r] = GO[tid + 256 * r]; }
    };
    // a network input into LDS (and the barrier in front of eval)
    auto put_in = [&](const float (&h0)[NKD], const float (&h1)[NKD]) {
#pragma unroll
        for (int r = 0; r < NKD; ++r) { X0[tid + 256 * r] = h0[r]; X1[tid + 256 * r] = h1[r]; }
        __syncthreads();
    };

    const int tile = wg;                      // one workgroup per 16-column tile
    const bool tile_ok = true;
    const int gq = tid >> 4, gcol = tile * 16 + (tid & 15);      // feature of element r: gq + 16 r
    const bool colok = gcol < Q.B;
    const size_t fo = (size_t)tid;
    const double N = (double)Q.D * (double)Q.B;
    const float dtmax = Q.t1 - Q.t0;
    const float dtmin = 1.1920929e-7f;

    float up[NKD], dW[NKD], dZ[NKD];
#pragma unroll
    for (int q = 0; q < NKD; ++q) { up[q] = ldc(Q.x, Q.D, gcol, 16 * q + gq, colok); dW[q] = 0.f; dZ[q] = 0.f; }

    // draw `d` of the pool for this lane's elements
    auto xi = [&](int d, int wz, int q) -> float {
        const int f = 16 * q + gq;
        return (colok && f < Q.D) ? Q.noise[(((size_t)d * 2 + wz) * Q.B + gcol) * Q.D + f] : 0.f;
    };

    // ---- initial step size: sde_determine_initdt (StochasticDiffEq src/initdt.jl) ----
    float dt;
    int seq = 0;
    {
        float f0[NKD], g0[NKD];
        put_in(up, up);
        eval(f0, g0);
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            g0[q] *= 3.f;
            if (colok && 16 * q + gq < Q.D) {
                const float sk = Q.abstol + fabsf(up[q]) * Q.reltol;
                const float a = up[q] / sk, b = fmaxf(fabsf(f0[q] + g0[q]), fabsf(f0[q] - g0[q])) / sk;
                pa += a * a; pb += b * b;
            }
        }
        pa = wave_sum_f(pa); pb = wave_sum_f(pb);
        if (lane == 0) { RED[wave] = pa; RED[kCW + wave] = pb; }
        __syncthreads();
        if (wave == 0) {
            float mine[2] = {0.f, 0.f};
            for (int w = 0; w < kCW; ++w) { mine[0] += RED[w]; mine[1] += RED[kCW + w]; }
            double o[2];
            const bool ok = sde_exchange<2>(Q, seq, mine, o, wg, lane);
            if (lane == 0) { DEC->xsum[0] = o[0]; DEC->xsum[1] = o[1]; DEC->status = ok ? 0 : 5; }
        }
        __syncthreads();
        ++seq;
        if (DEC->status) { if (wg == 0 && tid == 0) { SdeFinal F{}; F.status = 5; *Q.fin = F; } return; }
        const float d0 = (float)sqrt(DEC->xsum[0] / N), d1 = (float)sqrt(DEC->xsum[1] / N);
        float dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : (d0 / d1) / 100.f;
        if (dtmax < dt0) dt0 = dtmax;
        float u1[NKD], f1[NKD], g1[NKD];
#pragma unroll
        for (int q = 0; q < NKD; ++q) u1[q] = up[q] + dt0 * f0[q];
        put_in(u1, u1);
        eval(f1, g1);
        float pc = 0.f;
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            g1[q] *= 3.f;
            if (colok && 16 * q + gq < Q.D) {
                const float sk = Q.abstol + fabsf(up[q]) * Q.reltol;
                const float dg = fmaxf(fabsf(g0[q] - g1[q]), fabsf(g0[q] + g1[q]));
                const float c = fmaxf(fabsf(f1[q] - f0[q] + dg), fabsf(f1[q] - f0[q] - dg)) / sk;
                pc += c * c;
            }
        }
        __syncthreads();   // RED reuse
        pc = wave_sum_f(pc);
        if (lane == 0) RED[wave] = pc;
        __syncthreads();
        if (wave == 0) {
            float mine[1] = {0.f};
            for (int w = 0; w < kCW; ++w) mine[0] += RED[w];
            double o[1];
            const bool ok = sde_exchange<1>(Q, seq, mine, o, wg, lane);
            if (lane == 0) { DEC->xsum[0] = o[0]; DEC->status = ok ? 0 : 5; }
        }
        __syncthreads();
        ++seq;
        if (DEC->status) { if (wg == 0 && tid == 0) { SdeFinal F{}; F.status = 5; *Q.fin = F; } return; }
        const float d2 = (float)sqrt(DEC->xsum[0] / N) / dt0;
        const float m = d1 > d2 ? d1 : d2;
        float dt1;
        if (m <= 1e-15f) dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
        else dt1 = (float)pow(10.0, (double)(-(2.f + log10f(m)) / (Q.order + 0.5f)));
        dt = 100.f * dt0;
        if (dt1 < dt) dt = dt1;
        if (dtmax < dt) dt = dtmax;
        if (Q.replay) dt = Q.replay[0];
    }
    float t = Q.t0, qold = Q.qoldinit;
    if (Q.t1 - t < dt) dt = Q.t1 - t;
    // first increments: draw 0, one piece of the current step
    int my_status = 0;
    if (Q.n_pool < 1) my_status = 4;
    {
        const float s = sqrtf(fabsf(dt));
#pragma unroll
        for (int q = 0; q < NKD; ++q) { dW[q] = s * xi(0, 0, q); dZ[q] = s * xi(0, 1, q); }
        if (tile_ok) {
            float* w0 = Q.slots + sde_slot_off(Q, 0, 0, tile, 8) + fo;
            float* z0 = Q.slots + sde_slot_off(Q, 0, 1, tile, 8) + fo;
#pragma unroll
            for (int q = 0; q < NKD; ++q) { w0[q * 256] = dW[q]; z0[q * 256] = dZ[q]; }
        }
        if (tid == 0) { STK->next_slot = 1; STK->next_draw = 1; STK->n2 = 1; S2L[0] = dt; S2s[0] = 0; STK->Wdt = dt; }
    }
    int n = 0, n_acc = 0, next_save = 0;
    if (Q.nsave > 0 && Q.sv_t[0] == Q.t0) {      // save_start: t0 itself is a save time
#pragma unroll
        for (int q = 0; q < NKD; ++q) if (colok && 16 * q + gq < Q.D) Q.sv_out[((size_t)gcol * Q.nsave) * Q.D + 16 * q + gq] = up[q];
        next_save = 1;
    }
    __syncthreads();

    // ---- the solve ----
    int next_draw_known = 1;          // (draw 0 went into the first step's increments above)
    while (true) {
        // loop-top checks (identical in every thread of every workgroup)
        int status = my_status;
        bool stop = false;
        if (status == 0) {
            if (!(t < Q.t1) || (Q.replay && n >= Q.n_replay)) stop = true;
            else if (n >= Q.max_attempts) { status = 1; stop = true; }
            else if (dt != dt) { status = 3; stop = true; }
            else if (!(dt > dtmin)) { status = 2; stop = true; }
        } else stop = true;
        if (stop) { my_status = status; break; }

        // the pool's next draw -- the one an accept or a rejection's bridge consumes after this attempt's meeting -- is requested now: read at
        // the point of use it was a cold load at the end of every attempt, on the way to the next one
        const int pf_draw = next_draw_known < Q.n_pool ? next_draw_known : -1;
        float pxi[2][NKD];
#pragma unroll
        for (int q = 0; q < NKD; ++q) { pxi[0][q] = pf_draw >= 0 ? xi(pf_draw, 0, q) : 0.f; pxi[1][q] = pf_draw >= 0 ? xi(pf_draw, 1, q) : 0.f; }
        auto xin = [&](int dr, int wz, int q) -> float { return dr == pf_draw ? pxi[wz][q] : xi(dr, wz, q); };

        const float sqdt = sqrtf(fabsf(dt));
        float k[4][NKD], g[4][NKD], un[NKD];
        float part = 0.f, eig0 = 0.f, eig1 = 0.f;
        {   // one attempted SRI step (sde_attempt of rnde_sde.h on this thread's elements; the network evaluations through LDS)
            const SriTableau& T = Q.T;
            float chi2[NKD], hd[NKD];      // hd: H0_4 - H0_3 (reg_kind 2: the denominator of the stiffness estimate)
            const float sqrt3 = 1.7320508075688772f;
#pragma unroll
            for (int q = 0; q < NKD; ++q) { chi2[q] = (dW[q] + dZ[q] / sqrt3) / 2.f; hd[q] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float h0[NKD], h1[NKD];
#pragma unroll
                for (int q = 0; q < NKD; ++q) {
                    float a0 = 0.f, b0 = 0.f, a1_ = 0.f, b1_ = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < s) {
                            a0 += T.A0[4 * s + j] * k[j][q]; b0 += T.B0[4 * s + j] * g[j][q];
                            a1_ += T.A1[4 * s + j] * k[j][q]; b1_ += T.B1[4 * s + j] * g[j][q];
                        }
                    h0[q] = s ? up[q] + dt * a0 + chi2[q] * b0 : up[q];
                    h1[q] = s ? up[q] + dt * a1_ + sqdt * b1_ : up[q];
                    if (Q.reg_kind == 2) { if (s == 2) hd[q] = h0[q]; if (s == 3) hd[q] = h0[q] - hd[q]; }
                }
                put_in(h0, h1);
                eval(k[s], g[s]);
            }
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const float w = dW[q];
                const float chi1 = (w * w - fabsf(dt)) / (2.f * sqdt);
                const float chi3 = (w * w * w - 3.f * w * dt) / (6.f * dt);
                float sa = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, sk_ = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sa += T.alpha[j] * k[j][q]; sk_ += k[j][q];
                    s1 += T.beta1[j] * g[j][q]; s2 += T.beta2[j] * g[j][q];
                    s3 += T.beta3[j] * g[j][q]; s4 += T.beta4[j] * g[j][q];
                }
                const float E2 = chi2[q] * s3 + chi3 * s4;
                const float u = up[q] + dt * sa + E2 + w * s1 + chi1 * s2;
                un[q] = u;
                if (colok && 16 * q + gq < Q.D) {
                    const float E1 = dt * sk_;
                    const float sc = Q.abstol + fmaxf(fabsf(up[q]), fabsf(u)) * Q.reltol;
                    const float r = (Q.delta * E1 + E2) / sc;
                    part += r * r;
                    if (Q.reg_kind == 2) { const float v1 = k[3][q] - k[2][q]; eig0 += v1 * v1; eig1 += hd[q] * hd[q]; }
                }
            }
        }
        part = wave_sum_f(part);
        if (Q.reg_kind == 2) { eig0 = wave_sum_f(eig0); eig1 = wave_sum_f(eig1); }
        if (lane == 0) { RED[wave] = part; if (Q.reg_kind == 2) { RED[2 * kCW + wave] = eig0; RED[3 * kCW + wave] = eig1; } }
        __syncthreads();
        if (wave == 0) {
            float mine[1] = {0.f};
            for (int w = 0; w < kCW; ++w) mine[0] += RED[w];
            if (Q.reg_kind == 2 && lane == 0) {      // this workgroup's share of the two norms of attempt n (summed behind the solve: rnde_sde_eig_reduce_kernel)
                float e0 = 0.f, e1 = 0.f;
                for (int w = 0; w < kCW; ++w) { e0 += RED[2 * kCW + w]; e1 += RED[3 * kCW + w]; }
                Q.eigpart[((size_t)n * 2) * Q.nwg + wg] = e0; Q.eigpart[((size_t)n * 2 + 1) * Q.nwg + wg] = e1;
            }
            double o[1];
            const bool ok = sde_exchange<1>(Q, seq, mine, o, wg, lane);
            if (lane == 0) {
                const SdeCtlView V{t, dt, qold, dtmax, dtmin, n, n_acc, next_save, cap, S1L, S1s, S2L, S2s, FREEL, STK, OPS, DEC};
                sde_decide(Q, V, ok, o[0], N, wg);
            }
        }
        __syncthreads();
        ++seq;
        const SdeDecision d = *DEC;
        next_draw_known = d.n_draws;
        if (d.status) { my_status = d.status; ++n; break; }
        if (d.accepted) {
            if (Q.keep_tape && tile_ok) {
                float* R = Q.tape + ((size_t)n_acc * 12 * Q.ntiles + tile) * 512 + fo;
                const size_t as = (size_t)Q.ntiles * 512;
#pragma unroll
                for (int q = 0; q < NKD; ++q) {
                    R[q * 256] = up[q]; R[as + q * 256] = dW[q]; R[2 * as + q * 256] = dZ[q];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { R[(3 + j) * as + q * 256] = k[j][q]; R[(7 + j) * as + q * 256] = g[j][q]; }
                    R[11 * as + q * 256] = un[q];
                }
            }
            for (int idx = d.sv_lo; idx < d.sv_hi; ++idx) {      // saveat: linear interpolant of the SDE solution inside the step
                const float tsv = Q.sv_t[idx];
                const bool at_end = (tsv == d.t);
                const float th = (tsv - t) / dt;
#pragma unroll
                for (int q = 0; q < NKD; ++q)
                    if (colok && 16 * q + gq < Q.D) Q.sv_out[((size_t)gcol * Q.nsave + idx) * Q.D + 16 * q + gq] = at_end ? un[q] : (1.f - th) * up[q] + th * un[q];
            }
            next_save = d.sv_hi;
#pragma unroll
            for (int q = 0; q < NKD; ++q) up[q] = un[q];
            ++n_acc;
            // array operations of accept_step!
            float aw[NKD], az[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) { aw[q] = 0.f; az[q] = 0.f; }
            for (int i = 0; i < d.nops; ++i) {
                const SdeOp op = OPS[i];
                float* pw = Q.slots + sde_slot_off(Q, op.a, 0, tile, 8) + fo;
                float* pz = Q.slots + sde_slot_off(Q, op.a, 1, tile, 8) + fo;
                if (op.type == OP_ADD) {
#pragma unroll
                    for (int q = 0; q < NKD; ++q) if (tile_ok) { aw[q] += pw[q * 256]; az[q] += pz[q * 256]; }
                } else if (op.type == OP_BRIDGE) {
                    float* nw = Q.slots + sde_slot_off(Q, op.b, 0, tile, 8) + fo;
                    float* nz = Q.slots + sde_slot_off(Q, op.b, 1, tile, 8) + fo;
#pragma unroll
                    for (int q = 0; q < NKD; ++q) if (tile_ok) {
                        const float lw = pw[q * 256], lz = pz[q * 256];
                        const float bw = op.f0 * lw + op.f1 * xin(op.draw, 0, q), bz = op.f0 * lz + op.f1 * xin(op.draw, 1, q);
                        aw[q] += bw; az[q] += bz;
                        if (op.flags & 1) { pw[q * 256] = lw - bw; pz[q * 256] = lz - bz; }
                        if (op.flags & 2) { nw[q * 256] = bw; nz[q * 256] = bz; }
                    }
                } else if (op.type == OP_FRESH) {
#pragma unroll
                    for (int q = 0; q < NKD; ++q) if (tile_ok) {
                        const float fw = op.f0 * xin(op.draw, 0, q), fz = op.f0 * xin(op.draw, 1, q);
                        aw[q] += fw; az[q] += fz;
                        pw[q * 256] = fw; pz[q * 256] = fz;
                    }
                }
            }
            if (!d.done) {
#pragma unroll
                for (int q = 0; q < NKD; ++q) { dW[q] = aw[q]; dZ[q] = az[q]; }
            }
            qold = d.eest > Q.qoldinit ? d.eest : Q.qoldinit;
        } else if (!d.done) {
            float tw[NKD], tz[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) { tw[q] = 0.f; tz[q] = 0.f; }
            for (int i = 0; i < d.nops; ++i) {
                const SdeOp op = OPS[i];
                float* pw = Q.slots + sde_slot_off(Q, op.a, 0, tile, 8) + fo;
                float* pz = Q.slots + sde_slot_off(Q, op.a, 1, tile, 8) + fo;
                if (op.type == OP_SUB) {
#pragma unroll
                    for (int q = 0; q < NKD; ++q) if (tile_ok) { tw[q] += pw[q * 256]; tz[q] += pz[q * 256]; }
                } else if (op.type == OP_RBRIDGE) {
                    float* cw = Q.slots + sde_slot_off(Q, op.b, 0, tile, 8) + fo;
                    float* cz = Q.slots + sde_slot_off(Q, op.b, 1, tile, 8) + fo;
#pragma unroll
                    for (int q = 0; q < NKD; ++q) {
                        const float K2 = dW[q] - tw[q], K3 = dZ[q] - tz[q];
                        const float bw = op.f0 * K2 + op.f1 * xin(op.draw, 0, q), bz = op.f0 * K3 + op.f1 * xin(op.draw, 1, q);
                        if (tile_ok) {
                            if (op.flags & 1) { pw[q * 256] = K2 - bw; pz[q * 256] = K3 - bz; }
                            cw[q * 256] = bw; cz[q * 256] = bz;
                        }
                        dW[q] = bw; dZ[q] = bz;
                    }
                }
            }
        }
        t = d.t; dt = d.dt;
        ++n;
        __syncthreads();   // DEC / OPS / RED are rewritten by the next attempt
        if (d.done) break;
    }
    if (Q.u_out) {
#pragma unroll
        for (int q = 0; q < NKD; ++q) if (colok && 16 * q + gq < Q.D) Q.u_out[(size_t)gcol * Q.D + 16 * q + gq] = up[q];
    }
    if (wg == 0 && tid == 0) { SdeFinal F{n, n_acc, my_status, STK->next_draw, t, 0.f, 0.f, 0.f}; *Q.fin = F; }
}


// ---- reverse sweep, four waves per tile: one launch, no meeting (see rnde_sde_bwd_kernel).  Per stage, backwards: element-wise part
// (layer inputs / outputs and the output layers' pre-activation cotangents to the slab and LDS), phase 1 = every wave recomputes its
// hidden tile of the drift AND forms (W2^T z2) for the same tile -- both land in the same D registers, so z1 = (W2^T z2) (1 - h^2)
// needs no exchange --, phase 2 = W1^T z1 on waves 0, 1 beside Wg^T zg on waves 2, 3.  Three barriers per stage; the slab rows are
// written exactly as rnde_sde_bwd_kernel writes them (rnde_chain_wgrad_kernel contracts them afterwards).
constexpr int kSmwBwdLdsFloats = 3584;   // X0, Z2, ZG, Z1, HBF, HBG
__global__ __launch_bounds__(kSmwThreads) void rnde_sde_bwd_mw_kernel(const SdeBwdParams Bq) {
    constexpr int NKD = 2;      // element e = tid + 256 r <-> (feature e >> 4, column e & 15)
    const SdeParams& Q = Bq.F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* X0 = smem;                 // [32][16] drift input h0 of the stage (layer 1 is recomputed)
    float* Z2 = smem + 512;           // [32][16] cotangent of the drift's pre-activation output
    float* ZG = smem + 1024;          // [32][16] the diffusion's
    float* Z1 = smem + 1536;          // [64][16] cotangent of the drift's hidden pre-activation
    float* HBF = smem + 2560;         // [32][16] cotangent of h0
    float* HBG = smem + 3072;         // [32][16] cotangent of h1
    const int tile = blockIdx.x;
    const int lg = lane >> 4, lc = lane & 15;
    const int gq = tid >> 4, gcol = tile * 16 + (tid & 15);
    const bool colok = gcol < Q.B;
    // register-stationary fragments of this wave (packed tables of the one-wave engine, see rnde_sdemw.h's forward kernel):
    // drift layer 1 tile `wave` (forward, recomputed), W2^T tile `wave` (32 -> 64), then W1^T tile wave (waves 0, 1) or Wg^T tile wave - 2
    float a1[8], b1[4], t2[8], t1[16];
    {
        const float* TFf = Q.frags_f + (size_t)(Q.Gf.nfrag_f + Q.Gf.nfrag_b) * 64;
        const float* TFg = Q.frags_g + (size_t)(Q.Gg.nfrag_f + Q.Gg.nfrag_b) * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = Q.frags_f[((size_t)Q.Gf.foff[0] + wave * 8 + k) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) b1[i] = Q.frags_f[((size_t)Q.Gf.nfrag_f + Q.Gf.boff[0] + 4 * wave + i) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 8; ++k) t2[k] = TFf[((size_t)Q.Gf.toff[1] + wave * 8 + k) * 64 + lane];
        if (wave < 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t1[k] = TFf[((size_t)Q.Gf.toff[0] + wave * 16 + k) * 64 + lane];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) t1[k] = k < 8 ? TFg[((size_t)Q.Gg.toff[0] + (wave - 2) * 8 + (k < 8 ? k : 0)) * 64 + lane] : 0.f;
        }
    }
    const bool th1 = Q.Gf.act[0] != 0, th2 = Q.Gf.act[1] != 0, thg = Q.Gg.act[0] != 0;
    const SriTableau& T = Q.T;
    const BChainParams& Cf = Bq.Cf;   // the two nets as chain_fbwd wants them (kernel-argument memory: their tables stay scalar loads)
    const BChainParams& Cg = Bq.Cg;
    const double N = (double)Q.D * (double)Q.B;
    const float sqrt3 = 1.7320508075688772f;
    const size_t as = (size_t)Q.ntiles * 512;
    float U[NKD];
#pragma unroll
    for (int q = 0; q < NKD; ++q) U[q] = Bq.nsave > 0 ? 0.f : ldc(Bq.ubar, Q.D, gcol, 16 * q + gq, colok);
    // The tape record of a step (12 arrays: uprev, dW, dZ, k1..4, g1..4, unew) is requested one step AHEAD: each iteration used to begin with
    // 24 cold loads per thread and their wait (~1.5 us of a ~7.5 us step)
    float rec[12][NKD];
    auto request = [&](int a_) {
        const float* Rr = Q.tape + ((size_t)a_ * 12 * Q.ntiles + tile) * 512 + tid;
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int q = 0; q < NKD; ++q) rec[j][q] = __builtin_nontemporal_load(Rr + (size_t)j * as + q * 256);
    };
    if (Bq.n_acc > 0) request(Bq.n_acc - 1);
    for (int a = Bq.n_acc - 1; a >= 0; --a) {
        const SdeMeta m = Bq.acc_meta[a];
        const float dt = m.dt, sqdt = sqrtf(fabsf(dt));
        float up[NKD], dW[NKD], dZ[NKD], k[4][NKD], g[4][NKD], kb[4][NKD], gb[4][NKD], upb[NKD], chi2[NKD], unw[NKD];
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            up[q] = rec[0][q]; dW[q] = rec[1][q]; dZ[q] = rec[2][q]; unw[q] = rec[11][q];
#pragma unroll
            for (int j = 0; j < 4; ++j) { k[j][q] = rec[3 + j][q]; g[j][q] = rec[7 + j][q]; }
        }
        if (a > 0) request(a - 1);
        float svup[NKD];
#pragma unroll
        for (int q = 0; q < NKD; ++q) svup[q] = 0.f;
        for (int idx = m.sv_lo; idx < m.sv_hi && Bq.nsave > 0; ++idx) {      // saveat points of this step: u(ts) = (1 - th) uprev + th u
            const float tsv = Bq.sv_t[idx];
            const bool at_end = (tsv == m.t + dt);
            const float th = at_end ? 1.f : (tsv - m.t) / dt;
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const float ub = (colok && 16 * q + gq < Q.D) ? Bq.ubar[((size_t)gcol * Bq.nsave + idx) * Q.D + 16 * q + gq] : 0.f;
                U[q] += th * ub; svup[q] += (1.f - th) * ub;
            }
        }
        const double eb = (Q.reg_kind == 1) ? (double)Bq.svb_acc[a] * (double)dt : 0.0;   // saveval = EEst * dt, dt constant
        const float coef = m.eest > 0.f ? (float)(eb / (N * (double)m.eest)) : 0.f;
        float c1 = 0.f, c2 = 0.f, v2[NKD];      // reg_kind 2: the stiffness estimate's cotangents (see rnde_sde_bwd_kernel)
        if (Q.reg_kind == 2 && m.n1 > 0.f && m.n2 > 0.f) {
            const double eigb = (double)Bq.svb_acc[a] / (double)Q.stab;
            c1 = (float)(eigb / (N * (double)m.n1 * (double)m.n2));
            c2 = (float)(-eigb * (double)m.n1 / (N * (double)m.n2 * (double)m.n2 * (double)m.n2));
        }
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            const float un = unw[q];
            const float w = dW[q];
            const float chi1 = (w * w - fabsf(dt)) / (2.f * sqdt);
            chi2[q] = (w + dZ[q] / sqrt3) / 2.f;
            const float chi3 = (w * w * w - 3.f * w * dt) / (6.f * dt);
            float sk_ = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { sk_ += k[j][q]; s3 += T.beta3[j] * g[j][q]; s4 += T.beta4[j] * g[j][q]; }
            float unb = U[q], upv = 0.f, numb = 0.f;
            if (colok && 16 * q + gq < Q.D) {
                const float E2 = chi2[q] * s3 + chi3 * s4, E1 = dt * sk_;
                const float au = fabsf(up[q]), an = fabsf(un);
                const bool use_new = !(au > an);
                const float sc = Q.abstol + (use_new ? an : au) * Q.reltol;
                const float res = (Q.delta * E1 + E2) / sc;
                const float rb = coef * res;
                numb = rb / sc;
                const float scb = -rb * res / sc;
                if (use_new) unb += scb * Q.reltol * sgnf(un); else upv += scb * Q.reltol * sgnf(up[q]);
            }
            upv += unb;
            const float e2b = unb + numb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kb[j][q] = dt * T.alpha[j] * unb + dt * Q.delta * numb;
                gb[j][q] = (w * T.beta1[j] + chi1 * T.beta2[j]) * unb + (chi2[q] * T.beta3[j] + chi3 * T.beta4[j]) * e2b;
            }
            upb[q] = upv + svup[q];
            v2[q] = 0.f;
            if (c1 != 0.f && colok && 16 * q + gq < Q.D) {
                const float v1 = k[3][q] - k[2][q];
                kb[3][q] += c1 * v1; kb[2][q] -= c1 * v1;
                float a3 = 0.f, b3 = 0.f, a2 = 0.f, b2 = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) { a3 += T.A0[12 + j] * k[j][q]; b3 += T.B0[12 + j] * g[j][q]; if (j < 2) { a2 += T.A0[8 + j] * k[j][q]; b2 += T.B0[8 + j] * g[j][q]; } }
                v2[q] = (up[q] + dt * a3 + chi2[q] * b3) - (up[q] + dt * a2 + chi2[q] * b2);
            }
        }
#pragma unroll
        for (int s = 3; s >= 0; --s) {
            float h0[NKD], h1[NKD], hb[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < s) {
                        a0 += T.A0[4 * s + j] * k[j][q]; b0 += T.B0[4 * s + j] * g[j][q];
                        a1 += T.A1[4 * s + j] * k[j][q]; b1 += T.B1[4 * s + j] * g[j][q];
                    }
                h0[q] = s ? up[q] + dt * a0 + chi2[q] * b0 : up[q];
                h1[q] = s ? up[q] + dt * a1 + sqdt * b1 : up[q];
            }
            float* slf = Cf.slab + (size_t)(4 * a + s) * Cf.ev_stride + ((size_t)tile * Cf.RS) * 64;
            float* slg = Cg.slab + (size_t)(4 * a + s) * Cg.ev_stride + ((size_t)tile * Cg.RS) * 64;
            // element-wise: layer inputs / outputs to the slab (what the parameter-gradient kernel contracts), pre-activation cotangents of
            // both output layers to LDS and slab, h0 to LDS for the recomputation of the hidden layer
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const int e = tid + 256 * q;
                const float ko = k[s][q], go = g[s][q];
                const float z2 = th2 ? kb[s][q] * (1.f - ko * ko) : kb[s][q];
                const float zg = thg ? gb[s][q] * (1.f - go * go) : gb[s][q];
                X0[e] = h0[q]; Z2[e] = z2; ZG[e] = zg;
                slf[(size_t)Cf.hrow[0] * 64 + e] = h0[q]; slf[(size_t)Cf.hrow[2] * 64 + e] = ko; slf[(size_t)Cf.zrow[1] * 64 + e] = z2;
                slg[(size_t)Cg.hrow[0] * 64 + e] = h1[q]; slg[(size_t)Cg.hrow[1] * 64 + e] = go; slg[(size_t)Cg.zrow[0] * 64 + e] = zg;
            }
            __syncthreads();
            {   // phase 1, wave w: hidden tile w recomputed, (W2^T z2) tile w, their product -> Z1
                float bv[8], zv[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { bv[kk] = X0[64 * kk + lane]; zv[kk] = Z2[64 * kk + lane]; }
                f32x4 acc0 = {b1[0], b1[1], b1[2], b1[3]}, acc1 = {0.f, 0.f, 0.f, 0.f}, ab0 = {0.f, 0.f, 0.f, 0.f}, ab1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 8; kk += 2) {
                    acc0 = mfma16(a1[kk], bv[kk], acc0); acc1 = mfma16(a1[kk + 1], bv[kk + 1], acc1);
                    ab0 = mfma16(t2[kk], zv[kk], ab0); ab1 = mfma16(t2[kk + 1], zv[kk + 1], ab1);
                }
                f32x4 hm = acc0 + acc1;
                if (th1) { const f32x2 t01 = tanh_fast2((f32x2){hm[0], hm[1]}), t23 = tanh_fast2((f32x2){hm[2], hm[3]}); hm = (f32x4){t01.x, t01.y, t23.x, t23.y}; }
                const f32x4 ab = ab0 + ab1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = (16 * wave + 4 * i + lg) * 16 + lc;
                    const float z1 = th1 ? ab[i] * (1.f - hm[i] * hm[i]) : ab[i];
                    Z1[e] = z1;
                    slf[(size_t)Cf.hrow[1] * 64 + e] = hm[i];
                    slf[(size_t)Cf.zrow[0] * 64 + e] = z1;
                }
            }
            __syncthreads();
            {   // phase 2: waves 0, 1 = (W1^T z1) tiles (16 k-steps), waves 2, 3 = (Wg^T zg) tiles (8 k-steps)
                const float* zb = (wave < 2 ? Z1 : ZG) + lane;
                float zv[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) zv[kk] = (kk < 8 || wave < 2) ? zb[64 * ((kk < 8 || wave < 2) ? kk : 0)] : 0.f;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 8; kk += 2) { acc0 = mfma16(t1[kk], zv[kk], acc0); acc1 = mfma16(t1[kk + 1], zv[kk + 1], acc1); }
                if (wave < 2) {
#pragma unroll
                    for (int kk = 8; kk < 16; kk += 2) { acc0 = mfma16(t1[kk], zv[kk], acc0); acc1 = mfma16(t1[kk + 1], zv[kk + 1], acc1); }
                }
                const f32x4 o = acc0 + acc1;
                float* dst = wave < 2 ? HBF : HBG;
                const int mo = wave & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(16 * mo + 4 * i + lg) * 16 + lc] = o[i];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const float hbf = HBF[tid + 256 * q] + (s >= 2 ? (s == 3 ? c2 : -c2) * v2[q] : 0.f), hbg = HBG[tid + 256 * q];
                upb[q] += hbf;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < s) { kb[j][q] += dt * T.A0[4 * s + j] * hbf; gb[j][q] += chi2[q] * T.B0[4 * s + j] * hbf; }
                upb[q] += hbg;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < s) { kb[j][q] += dt * T.A1[4 * s + j] * hbg; gb[j][q] += sqdt * T.B1[4 * s + j] * hbg; }
            }
        }
#pragma unroll
        for (int q = 0; q < NKD; ++q) U[q] = upb[q];
    }
#pragma unroll
    for (int q = 0; q < NKD; ++q)
        if (colok && 16 * q + gq < Q.D) {
            float v = U[q];
            if (Bq.nsave > 0 && Bq.save_t0) v += Bq.ubar[((size_t)gcol * Bq.nsave) * Q.D + 16 * q + gq];
            Bq.xbar[(size_t)gcol * Q.D + 16 * q + gq] = v;
        }
}

}  // namespace rnde
