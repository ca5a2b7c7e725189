// rnde_chain.h -- "chain engine": the adaptive Tsit5 attempt for SMALL-WIDTH Dense chains (every layer width <= 64),
// e.g. the latent-ODE dynamics of reference experiments/latent_ode.jl:113-124 (tanh, then 8 x Dense(20<->50, tanh),
// SURVEY.md 8d config 4) and the small two-layer networks of test/test_node.jl.
//
// Why a separate engine: at D = 20 the whole state of 16 batch columns is 5 VGPRs per array, so nothing about the
// MNIST engines (HBM-resident stage arrays, streamed weights) applies.  Here
//   * one WAVE owns 16 batch columns for the whole attempt: uprev, k1..k7, the running stage combinations and the
//     error estimate never leave its registers; nothing but the tape record is written;
//   * a Dense layer is a chain of v_mfma_f32_16x16x4_f32 whose B operand IS the previous layer's D registers: the
//     D fragment (reg i, lane group g) holds feature 4*ks + g of the column lane & 15, and the A fragments are
//     packed with the matching K permutation, so activations are never moved, transposed or staged through LDS;
//   * all weight fragments (<= ~150 KB) are copied into LDS once per launch and read with conflict-free
//     ds_read_b32 (64 consecutive floats per fragment);
//   * one launch per attempted step; the controller runs in the prologue exactly as in rnde_fwd.h (advance_state),
//     so the host loop, the tape metadata and the saveat bookkeeping are shared with the other engines.
//
// Layout vocabulary: an activation vector of width n is nks = ceil(n/4) "k-steps"; k-step ks of a wave is one VGPR
// whose lane (g = lane >> 4, col = lane & 15) holds feature 4*ks + g of batch column 16*tile + col (0 past n).
// Arrays in the arena use the same order: [tile][ks][lane] ("fragment order", 256-byte coalesced rows), with the
// k-step count padded to the kernel's compile-time NKD (zeros), and the bias tables / reverse-pass dumps padded to
// whole 16-feature tiles: every per-element loop in the kernels is then compile-time, and the only data-dependent
// control flow is one switch per layer on its tile counts (per-element uniform branches serialised each LDS read
// behind its own s_waitcnt: 260 cycles per MFMA measured, against 46 for the dependent chain itself).
// Caller arrays (x, u_out, the saveat output, cotangents) stay D x B column-major as the ABI says.
#pragma once
#include "rnde_fwd.h"
#include "rnde_stage.h"   // mfma16

namespace rnde {

constexpr int kCW = 4;          // waves per workgroup (16 batch columns each)
constexpr int kCMaxL = 8;       // == RNDE_MAX_LAYERS
constexpr int kCMaxKs = 16;     // k-steps of the widest activation (width <= 64)

struct ChainGeo {
    int n_layers, time_dep, pre_act, nksD;
    int width[kCMaxL + 1], nks[kCMaxL + 1];
    int act[kCMaxL];
    int poff[kCMaxL];                                  // offset of layer l inside the Flux.destructure vector
    int foff[kCMaxL], toff[kCMaxL], boff[kCMaxL];      // fragment offsets (units of 64 floats) in the three tables
    int nfrag_f, nfrag_b, nfrag_t;                     // forward A fragments | bias (+ time column) | transposed A fragments
};

struct ChainParams {
    StepParams F;          // shared controller / tape parameters (H and the packed-weight fields are unused)
    ChainGeo G;
    const float* frags;    // [nfrag_f + nfrag_b + nfrag_t][64]
    int ntiles;            // wave tiles: Bpad / 16
};

// record layout (fragment order arrays of ntiles * nksD * 64 floats): k2..k7 | unew | uprev copy | k1 copy | g2..g6
// (the stage inputs g_s are taped rather than recomputed so that the reverse pass linearises exactly the values the
//  forward evaluated)
struct ChainRec {
    long long A;
    int S = 7;          // stages of the pair in first-same-as-last form (Tsit5, DP5: 7; a pair that comes as a table: RkTab.S)
    __host__ __device__ long long k(int s) const { return (long long)(s - 2) * A; }   // s = 2..S
    __host__ __device__ long long unew() const { return (long long)(S - 1) * A; }
    __host__ __device__ long long upc() const { return (long long)S * A; }
    __host__ __device__ long long k1c() const { return (long long)(S + 1) * A; }
    __host__ __device__ long long g(int s) const { return (long long)(S + 2 + s - 2) * A; }   // s = 2..S-1
    __host__ __device__ long long total() const { return 2LL * S * A; }
};


// ---- packing: Flux.destructure vector -> fragment tables ------------------------------------------------------------
// forward fragment (l, mo, ks): lane = 16*kk + rho holds W_l[16 mo + 4 (rho & 3) + (rho >> 2)][4 ks + kk]
//   (M row rho of the MFMA = D register rho & 3 of lane group rho >> 2  <->  out feature 4 (4 mo + (rho & 3)) + (rho >> 2))
// transposed fragment (l, mi, ks): the same with the roles of the two widths exchanged (J^T products of the reverse pass)
// bias fragment (l, ks): lane group g holds b_l[4 ks + g]; followed, when time_dep, by the time column W_l[:, in]
static __global__ void rnde_chain_pack_kernel(const float* __restrict__ p, float* __restrict__ frags, const ChainGeo G) {
    const long long total = (long long)(G.nfrag_f + G.nfrag_b + G.nfrag_t) * 64;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        int fr = (int)(e >> 6);
        const int lane = (int)(e & 63), rho = lane & 15, kk = lane >> 4;
        float v = 0.f;
        if (fr < G.nfrag_f) {
            int l = 0;
            while (l + 1 < G.n_layers && fr >= G.foff[l + 1]) ++l;
            const int r = fr - G.foff[l], nin = G.nks[l], mo = r / nin, ks = r - mo * nin;
            const int fo = 16 * mo + 4 * (rho & 3) + (rho >> 2), fi = 4 * ks + kk;
            const int in = G.width[l], out = G.width[l + 1];
            if (fo < out && fi < in) v = p[G.poff[l] + (size_t)fi * out + fo];
        } else if (fr < G.nfrag_f + G.nfrag_b) {
            fr -= G.nfrag_f;
            int l = 0;
            while (l + 1 < G.n_layers && fr >= G.boff[l + 1]) ++l;
            int r = fr - G.boff[l];
            const int nout = 4 * ((G.nks[l + 1] + 3) >> 2), in = G.width[l], out = G.width[l + 1];   // padded to whole tiles
            const bool tcol = r >= nout;
            if (tcol) r -= nout;
            const int f = 4 * r + kk;
            if (f < out) v = tcol ? p[G.poff[l] + (size_t)in * out + f] : p[G.poff[l] + (size_t)(in + G.time_dep) * out + f];
        } else {
            fr -= G.nfrag_f + G.nfrag_b;
            int l = 0;
            while (l + 1 < G.n_layers && fr >= G.toff[l + 1]) ++l;
            const int r = fr - G.toff[l], nout = G.nks[l + 1], mi = r / nout, ks = r - mi * nout;
            const int fi = 16 * mi + 4 * (rho & 3) + (rho >> 2), fo = 4 * ks + kk;
            const int in = G.width[l], out = G.width[l + 1];
            if (fo < out && fi < in) v = p[G.poff[l] + (size_t)fi * out + fo];
        }
        frags[e] = v;
    }
}

// copy `units` KiB of fragment tables into LDS with the LDS-DMA path (no VGPR round trip, all loads in flight at once)
__device__ __forceinline__ void chain_fill_lds(const float* __restrict__ frags, float* smem, int units, int wave, int lane) {
    for (int u = wave; u < units; u += kCW) dma_unit((const f32x4*)(frags + (size_t)u * 256) + lane, smem + (size_t)u * 256);
    wait_vm<0>();
    __syncthreads();
}

// ---- acc[mo] += A(mo, .) * in   for the first `mt` output tiles; NKS = k-steps of the input --------------------------
// C chains of MFMAs, interleaved: chain c accumulates A fragments frag(c, k) for k in [0, n_c) into acc_c
template <int NKS>
__device__ __forceinline__ void chain_mm_t(const float* fr, int mt, const float (&in)[kCMaxKs], f32x4 (&acc)[4]) {
    // A dependent v_mfma_f32_16x16x4_f32 issues every 46 cycles, an independent one every 32 (tools/micro/mfma_chain):
    // keep up to 4 independent accumulation chains in flight -- the layer's output tiles, and when there are only 1-2 of
    // them the two halves of K (the half sums are added at the end; summation order differs from a serial dot product
    // by fp32 rounding only).  All A fragments of the layer are read from LDS in one batch first; the sched_barriers
    // keep the compiler from re-serialising every ds_read behind its own s_waitcnt.
    constexpr int K0 = (NKS + 1) / 2, K1 = NKS - K0;
    if (mt >= 3) {
        float a[4][NKS];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
            if (mo < 3 || mt == 4) {
#pragma unroll
                for (int k = 0; k < NKS; ++k) a[mo][k] = fr[(mo * NKS + k) * 64];
            }
        __builtin_amdgcn_sched_barrier(0);
        if (mt == 4) {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
#pragma unroll
                for (int mo = 0; mo < 4; ++mo) acc[mo] = mfma16(a[mo][k], in[k], acc[mo]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
#pragma unroll
                for (int mo = 0; mo < 3; ++mo) acc[mo] = mfma16(a[mo][k], in[k], acc[mo]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        float a[2][NKS];
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
            if (mo < mt) {
#pragma unroll
                for (int k = 0; k < NKS; ++k) a[mo][k] = fr[(mo * NKS + k) * 64];
            }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hi[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (mt == 2) {
#pragma unroll
            for (int k = 0; k < K0; ++k) {
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) {
                    acc[mo] = mfma16(a[mo][k], in[k], acc[mo]);
                    if (k < K1) hi[mo] = mfma16(a[mo][K0 + (k < K1 ? k : 0)], in[K0 + (k < K1 ? k : 0)], hi[mo]);
                }
            }
            acc[0] += hi[0]; acc[1] += hi[1];
        } else {
#pragma unroll
            for (int k = 0; k < K0; ++k) {
                acc[0] = mfma16(a[0][k], in[k], acc[0]);
                if (k < K1) hi[0] = mfma16(a[0][K0 + (k < K1 ? k : 0)], in[K0 + (k < K1 ? k : 0)], hi[0]);
            }
            acc[0] += hi[0];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void chain_mm(const float* fr, int nks, int mt, const float (&in)[kCMaxKs], f32x4 (&acc)[4]) {
    switch (nks) {
        case 1: chain_mm_t<1>(fr, mt, in, acc); break;   case 2: chain_mm_t<2>(fr, mt, in, acc); break;
        case 3: chain_mm_t<3>(fr, mt, in, acc); break;   case 4: chain_mm_t<4>(fr, mt, in, acc); break;
        case 5: chain_mm_t<5>(fr, mt, in, acc); break;   case 6: chain_mm_t<6>(fr, mt, in, acc); break;
        case 7: chain_mm_t<7>(fr, mt, in, acc); break;   case 8: chain_mm_t<8>(fr, mt, in, acc); break;
        case 9: chain_mm_t<9>(fr, mt, in, acc); break;   case 10: chain_mm_t<10>(fr, mt, in, acc); break;
        case 11: chain_mm_t<11>(fr, mt, in, acc); break; case 12: chain_mm_t<12>(fr, mt, in, acc); break;
        case 13: chain_mm_t<13>(fr, mt, in, acc); break; case 14: chain_mm_t<14>(fr, mt, in, acc); break;
        case 15: chain_mm_t<15>(fr, mt, in, acc); break; default: chain_mm_t<16>(fr, mt, in, acc); break;
    }
}

// accumulator init = bias (+ t * time column) for the MT output tiles of a layer; bf -> this lane's entry of row 0
template <int MT>
__device__ __forceinline__ void chain_bias_t(const float* bf, float ts, int td, f32x4 (&acc)[4]) {
    float b[4 * MT];
#pragma unroll
    for (int ks = 0; ks < 4 * MT; ++ks) b[ks] = bf[ks * 64];
    if (td) {
        float w[4 * MT];
#pragma unroll
        for (int ks = 0; ks < 4 * MT; ++ks) w[ks] = bf[(4 * MT + ks) * 64];
#pragma unroll
        for (int ks = 0; ks < 4 * MT; ++ks) b[ks] = fmaf(ts, w[ks], b[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc[ks >> 2][ks & 3] = ks < 4 * MT ? b[ks < 4 * MT ? ks : 0] : 0.f;
}
template <int MT>
__device__ __forceinline__ void chain_act_t(const f32x4 (&acc)[4], bool th, float (&a)[kCMaxKs]) {   // identity / generic
    if (th) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a[ks] = ks < 4 * MT ? tanh_fast(acc[ks >> 2][ks & 3]) : 0.f;
    } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a[ks] = ks < 4 * MT ? acc[ks >> 2][ks & 3] : 0.f;
    }
}
// tanh on exactly N k-steps, two per instruction (the rows past N are padding: zero)
template <int N>
__device__ __forceinline__ void chain_tanh_n(const f32x4 (&acc)[4], float (&a)[kCMaxKs]) {
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
        if (ks + 1 < N) {
            const f32x2 v = tanh_fast2((f32x2){acc[ks >> 2][ks & 3], acc[(ks + 1) >> 2][(ks + 1) & 3]});
            a[ks] = v.x; a[ks + 1] = v.y;
        } else if (ks < N) {
            a[ks] = tanh_fast(acc[ks >> 2][ks & 3]); a[ks + 1] = 0.f;
        } else { a[ks] = 0.f; a[ks + 1] = 0.f; }
    }
}
__device__ __forceinline__ void chain_tanh(const f32x4 (&acc)[4], int n, float (&a)[kCMaxKs]) {
    switch (n) {
        case 1: chain_tanh_n<1>(acc, a); break;   case 2: chain_tanh_n<2>(acc, a); break;   case 3: chain_tanh_n<3>(acc, a); break;
        case 4: chain_tanh_n<4>(acc, a); break;   case 5: chain_tanh_n<5>(acc, a); break;   case 6: chain_tanh_n<6>(acc, a); break;
        case 7: chain_tanh_n<7>(acc, a); break;   case 8: chain_tanh_n<8>(acc, a); break;   case 9: chain_tanh_n<9>(acc, a); break;
        case 10: chain_tanh_n<10>(acc, a); break; case 11: chain_tanh_n<11>(acc, a); break; case 12: chain_tanh_n<12>(acc, a); break;
        case 13: chain_tanh_n<13>(acc, a); break; case 14: chain_tanh_n<14>(acc, a); break; case 15: chain_tanh_n<15>(acc, a); break;
        default: chain_tanh_n<16>(acc, a); break;
    }
}

// one Dense layer on the wave's 16 columns: a <- act(W a + b (+ t * w_t)); all widths in k-steps.
// ALT = 0: any chain, shapes dispatched at run time.  The dispatch is not free: the structurised switches merge the
// 16-register accumulator / activation arrays after every case (measured: 32.5 k cycles per latent-ODE evaluation
// against 17.3 k with compile-time shapes, tools/micro/chain_eval.hip).  ALT = 1 therefore fixes the k-step pattern
// of the reference's own latent-ODE dynamics at compile time (experiments/latent_ode.jl:113-124: widths 20 <-> 50,
// i.e. 5 <-> 13 k-steps, alternating; depth, activations and flags stay run-time).
constexpr int kAltA = 5, kAltB = 13;
template <int ALT = 0>
__device__ __forceinline__ void chain_layer(const ChainGeo& G, const float* FR, const float* BF, int l, float ts, float (&a)[kCMaxKs], int lane, unsigned long long* dbg = nullptr) {
    const int nin = G.nks[l], mt = (G.nks[l + 1] + 3) >> 2;
    const float* bf = BF + (size_t)G.boff[l] * 64 + lane;
    f32x4 acc[4];
    if constexpr (ALT == 1) {
        constexpr int MA = (kAltA + 3) / 4, MB = (kAltB + 3) / 4;
        const bool th = G.act[l] != 0;
        if ((l & 1) == 0) {
            chain_bias_t<MB>(bf, ts, G.time_dep, acc);
            chain_mm_t<kAltA>(FR + (size_t)G.foff[l] * 64 + lane, MB, a, acc);
            if (th) chain_tanh_n<kAltB>(acc, a); else chain_act_t<MB>(acc, false, a);
        } else {
            chain_bias_t<MA>(bf, ts, G.time_dep, acc);
            chain_mm_t<kAltB>(FR + (size_t)G.foff[l] * 64 + lane, MA, a, acc);
            if (th) chain_tanh_n<kAltA>(acc, a); else chain_act_t<MA>(acc, false, a);
        }
        return;
    }
#ifdef RNDE_DIAG
    if (dbg) dbg[16 + 4 * l] = clock64();
#endif
#ifdef CH_NO_BIAS
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (false)
#endif
    switch (mt) {
        case 1: chain_bias_t<1>(bf, ts, G.time_dep, acc); break;
        case 2: chain_bias_t<2>(bf, ts, G.time_dep, acc); break;
        case 3: chain_bias_t<3>(bf, ts, G.time_dep, acc); break;
        default: chain_bias_t<4>(bf, ts, G.time_dep, acc); break;
    }
#ifdef RNDE_DIAG
    if (dbg) dbg[17 + 4 * l] = clock64();
#endif
#ifndef CH_NO_MFMA
    chain_mm(FR + (size_t)G.foff[l] * 64 + lane, nin, mt, a, acc);
#endif
#ifdef RNDE_DIAG
    if (dbg) dbg[18 + 4 * l] = clock64();
#endif
#ifdef CH_NO_TANH   // (ablation builds only: tools/micro/chain_eval.hip)
    if (false) {}
#else
    if (G.act[l] != 0) chain_tanh(acc, G.nks[l + 1], a);
#endif
    else switch (mt) {
        case 1: chain_act_t<1>(acc, false, a); break;
        case 2: chain_act_t<2>(acc, false, a); break;
        case 3: chain_act_t<3>(acc, false, a); break;
        default: chain_act_t<4>(acc, false, a); break;
    }
#ifdef RNDE_DIAG
    if (dbg) dbg[19 + 4 * l] = clock64();
#endif
}

// k = f(g, t) for the wave's columns (reference a6: dudt_, neural_ode.jl:55; latent_ode.jl:113-124)
template <int NKD, int ALT = 0>
__device__ __forceinline__ void chain_eval(const ChainGeo& G, const float* FR, const float* BF, float ts, const float (&g)[NKD], float (&out)[NKD], int lane, unsigned long long* dbg = nullptr) {
    float a[kCMaxKs];
#pragma unroll
    for (int k = 0; k < kCMaxKs; ++k) a[k] = (k < NKD) ? (G.pre_act ? tanh_fast(g[k < NKD ? k : 0]) : g[k < NKD ? k : 0]) : 0.f;
#pragma unroll 1
    for (int l = 0; l < G.n_layers; ++l) chain_layer<ALT>(G, FR, BF, l, ts, a, lane, dbg);
#pragma unroll
    for (int k = 0; k < NKD; ++k) out[k] = a[k];
}

// ---- element access --------------------------------------------------------------------------------------------------
// caller layout (D x B column-major) <-> k-step registers
__device__ __forceinline__ float ldc(const float* __restrict__ base, int D, int gcol, int f, bool ok) { return (ok && f < D) ? base[(size_t)gcol * D + f] : 0.f; }

enum { CM_STEP = 0, CM_INIT_A = 1, CM_INIT_B = 2, CM_FEVAL = 3, CM_FINISH = 4 };

// Dense output of the attempt accepted last (record Rp) at save indices [lo, hi), wave-tile local
template <int NKD>
__device__ __forceinline__ void chain_dense_points(const StepParams& P, const ChainRec& L, const float* Rp, size_t fo, int nksD, float tp, float dtp_,
                                                   float tnew, int lo, int hi, int gcol, int fb, int fs, bool colok) {
    float up[NKD], un[NKD], k[7][NKD];
#pragma unroll
    for (int q = 0; q < NKD; ++q) {
        if (q < nksD) {
            up[q] = Rp[L.upc() + fo + q * 64]; un[q] = Rp[L.unew() + fo + q * 64]; k[0][q] = Rp[L.k1c() + fo + q * 64];
#pragma unroll
            for (int j = 1; j < 7; ++j) k[j][q] = Rp[L.k(j + 1) + fo + q * 64];
        }
    }
    for (int idx = lo; idx < hi; ++idx) {
        const float ts = P.sv_t[idx];
        float b[7];
        const bool at_end = (ts == tnew);
        dense_weights((ts - tp) / dtp_, b);
#pragma unroll
        for (int q = 0; q < NKD; ++q) {
            if (q < nksD) {
                float o = un[q];
                if (!at_end) {
                    float acc = b[0] * k[0][q];
#pragma unroll
                    for (int j = 1; j < 7; ++j) acc += b[j] * k[j][q];
                    o = up[q] + dtp_ * acc;
                }
                const int f = fb + fs * q;
                if (colok && f < P.D) P.sv_out[((size_t)gcol * P.nsave + idx) * P.D + f] = o;
            }
        }
    }
}

#ifdef RNDE_DIAG
#define CHAIN_STAMP(i) do { if (MODE == CM_STEP && P.dbg_out && blockIdx.x == 0 && tid == 0) ((unsigned long long*)P.dbg_out)[i] = clock64(); } while (0)
#else
#define CHAIN_STAMP(i) do { } while (0)
#endif

template <int NKD, int MODE, int ALT = 0>
__global__ __launch_bounds__(64 * kCW) void rnde_chain_kernel(const ChainParams Q, const int n, float* __restrict__ u_out) {
    const StepParams& P = Q.F;
    const ChainGeo& G = Q.G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FR = smem;
    float* BF = FR + (size_t)G.nfrag_f * 64;
    const int fill_units = (G.nfrag_f + G.nfrag_b + 3) >> 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* RED = smem + (size_t)fill_units * 256;   // [3][kCW]
    const int tile = blockIdx.x * kCW + wave;
    const bool tile_ok = tile < Q.ntiles;
    // feature of register q of this lane = fb + fs * q; its batch column = gcol
    const int g = lane >> 4;
    const int fb = g, fs = 4;
    const int gcol = tile * 16 + (lane & 15);
    const bool colok = tile_ok && gcol < P.B;
    const bool writer = (blockIdx.x == 0 && tid == 0);
    constexpr int nksD = NKD;                                  // arena arrays are padded to NKD k-steps
    const ChainRec L{(long long)Q.ntiles * NKD * 64};
    const size_t fo = ((size_t)tile * NKD) * 64 + lane;        // fragment-order offset of this lane's k-step 0

    CHAIN_STAMP(0);
    if constexpr (MODE != CM_FINISH) chain_fill_lds(Q.frags, smem, fill_units, wave, lane);
    auto EVAL = [&](float ts_, const float (&gin)[NKD], float (&kout)[NKD]) { chain_eval<NKD, ALT>(G, FR, BF, ts_, gin, kout, lane); };
    CHAIN_STAMP(1);

    if constexpr (MODE == CM_FEVAL) {
        if (!tile_ok) return;
        float gv[NKD], kv[NKD];
#pragma unroll
        for (int q = 0; q < NKD; ++q) gv[q] = ldc(P.x, P.D, gcol, fb + fs * q, colok && q < nksD);
        EVAL(P.forced_t, gv, kv);
#pragma unroll
        for (int q = 0; q < NKD; ++q) if (q < nksD && colok && fb + fs * q < P.D) P.dbg_out[(size_t)gcol * P.D + fb + fs * q] = kv[q];
        return;
    } else if constexpr (MODE == CM_INIT_A || MODE == CM_INIT_B) {
        // ---- initial-step heuristic, SURVEY.md B.1 (same arithmetic as rnde_step_kernel) ----
        float dt0 = 0.f;
        if constexpr (MODE == CM_INIT_B) {
            const double N = (double)P.D * (double)P.Bn;
            const double s0 = sum_partials(P.initpart, P.nwg, lane);
            const double s1 = sum_partials(P.initpart + P.nwg, P.nwg, lane);
            const float d0 = (float)sqrt(s0 / N), d1 = (float)sqrt(s1 / N), dtmax = P.t1 - P.t0;
            int c0 = 0, cl = 0;
            if (d0 < 1e-5f || d1 < 1e-5f) { dt0 = 1e-6f; c0 = 1; }
            else dt0 = (d0 / d1) / 100.f;
            if (dtmax < dt0) { dt0 = dtmax; cl = 1; }
            if (writer) { P.initrec->d0 = d0; P.initrec->d1 = d1; P.initrec->dt0 = dt0; P.initrec->dt0_const = c0; P.initrec->dt0_clamped = cl; }
        }
        float pa = 0.f, pb = 0.f;
        if (tile_ok) {
            float xv[NKD], fv[NKD], gv[NKD], kv[NKD];
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                xv[q] = ldc(P.x, P.D, gcol, fb + fs * q, colok && q < nksD);
                fv[q] = 0.f;
                if constexpr (MODE == CM_INIT_B) {
                    if (q < nksD) fv[q] = P.f0[fo + q * 64];
                    gv[q] = xv[q] + dt0 * fv[q];
                    if (q < nksD) P.u1[fo + q * 64] = gv[q];
                } else gv[q] = xv[q];
            }
            EVAL((MODE == CM_INIT_B) ? P.t0 + dt0 : P.t0, gv, kv);
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                if (q < nksD) {
                    ((MODE == CM_INIT_B) ? P.f1 : P.f0)[fo + q * 64] = kv[q];
                    if (colok && fb + fs * q < P.D) {
                        const float sk = P.abstol + fabsf(xv[q]) * P.reltol;
                        if constexpr (MODE == CM_INIT_A) { const float a = xv[q] / sk, b = kv[q] / sk; pa += a * a; pb += b * b; }
                        else { const float a = (kv[q] - fv[q]) / sk; pa += a * a; }
                    }
                }
            }
        }
        pa = wave_sum_f(pa); pb = wave_sum_f(pb);
        if (lane == 0) { RED[wave] = pa; RED[kCW + wave] = pb; }
        __syncthreads();
        if (tid == 0) {
            float sa = 0.f, sb = 0.f;
            for (int w = 0; w < kCW; ++w) { sa += RED[w]; sb += RED[kCW + w]; }
            if constexpr (MODE == CM_INIT_A) { P.initpart[blockIdx.x] = sa; P.initpart[P.nwg + blockIdx.x] = sb; }
            else P.initpart[2 * P.nwg + blockIdx.x] = sa;
        }
        return;
    } else {
        // ---- controller, then (STEP) one attempted step / (FINISH) the copy-out ----
        const StepState S = advance_state(P, n, lane, writer, (MODE == CM_FINISH) ? P.ctl_final : &P.ctl[n & 1]);
        CHAIN_STAMP(2);
        if (P.nsave > 0 && tile_ok) {
            // saveat ({R,true} methods, neural_ode.jl:79-108): the points inside the step accepted last
            if (n == 0) {
                if (S.next_save > 0) {
#pragma unroll
                    for (int q = 0; q < NKD; ++q) {
                        const int f = fb + fs * q;
                        if (q < nksD && colok && f < P.D) P.sv_out[((size_t)gcol * P.nsave) * P.D + f] = P.x[(size_t)gcol * P.D + f];
                    }
                }
            } else {
                const StepState pv = P.ctl[(n - 1) & 1];
                const int lo = pv.next_save, hi = S.next_save;
                if (hi > lo && !pv.done) {
                    const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                    const float* Rp = P.arena + (long long)S.live * P.rec_stride;     // accepted => it is the live record
                    chain_dense_points<NKD>(P, L, Rp, fo, nksD, pv.t, dtp_, S.t, lo, hi, gcol, fb, fs, colok);
                }
            }
        }
        if constexpr (MODE == CM_FINISH) {
            if (!u_out || !tile_ok) return;
#pragma unroll
            for (int q = 0; q < NKD; ++q) {
                const int f = fb + fs * q;
                if (q < nksD && colok && f < P.D)
                    u_out[(size_t)gcol * P.D + f] = S.live < 0 ? P.x[(size_t)gcol * P.D + f] : P.arena[(long long)S.live * P.rec_stride + L.unew() + fo + q * 64];
            }
            return;
        } else {
            if (S.done) return;
            const float t = S.t;
            const float dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
            const int rec = P.tape ? n : (S.live == 0 ? 1 : 0);
            float* R = P.arena + (long long)rec * P.rec_stride;
            float part = 0.f, part1 = 0.f, part2 = 0.f;
            if (tile_ok) {
                // Rolled stage loop with shifting partial sums, exactly as rnde_step_kernel (rnde_fwd.h): Sa[i] is the
                // running combination sum_j a_{s+1+i,j} k_j of the i-th stage still to come; E = sum_j btilde_j k_j.
                float up[NKD], Sa[6][NKD], E[NKD], un[NKD], g6[NKD], k6[NKD];   // g6, k6: stiffness estimate only (reg_kind >= 2)
                const float* Rl = P.arena + (long long)(S.live < 0 ? 0 : S.live) * P.rec_stride;
#pragma unroll
                for (int q = 0; q < NKD; ++q) {
                    float k1 = 0.f;
                    up[q] = 0.f;
                    if (q < nksD) {
                        if (S.live < 0) { up[q] = ldc(P.x, P.D, gcol, fb + fs * q, colok); k1 = P.f0[fo + q * 64]; }
                        else { up[q] = Rl[L.unew() + fo + q * 64]; k1 = Rl[L.k(7) + fo + q * 64]; }
                        if (P.tape || P.nsave > 0) { R[L.upc() + fo + q * 64] = up[q]; R[L.k1c() + fo + q * 64] = k1; }
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) Sa[i][q] = kFwdShift[0][i] * k1;
                    E[q] = kTsBt[0] * k1;
                    un[q] = up[q]; g6[q] = 0.f; k6[q] = 0.f;
                }
                CHAIN_STAMP(3);
#pragma unroll 1
                for (int s = 1; s < 7; ++s) {   // zero-based stage: k_{s+1} = f(g_{s+1}, t + c_s dt)
                    float gq[NKD], kv[NKD];
#pragma unroll
                    for (int q = 0; q < NKD; ++q) gq[q] = up[q] + dt * Sa[0][q];
                    if (s == 6) {
#pragma unroll
                        for (int q = 0; q < NKD; ++q) { un[q] = gq[q]; if (q < nksD) R[L.unew() + fo + q * 64] = gq[q]; }
                    } else if (P.tape) {
#pragma unroll
                        for (int q = 0; q < NKD; ++q) if (q < nksD) R[L.g(s + 1) + fo + q * 64] = gq[q];
                    }
                    EVAL(t + kTsC[s] * dt, gq, kv);
                    CHAIN_STAMP(3 + s);
                    if (s == 5 && P.reg_kind >= 2) {
#pragma unroll
                        for (int q = 0; q < NKD; ++q) { g6[q] = gq[q]; k6[q] = kv[q]; }
                    }
                    if (s == 6 && P.reg_kind >= 2) {   // ||k7 - k6||^2, ||unew - g6||^2 (eigen_est of the composite algorithm)
#pragma unroll
                        for (int q = 0; q < NKD; ++q) {
                            if (colok && fb + fs * q < P.D) { const float d1 = kv[q] - k6[q], d2 = un[q] - g6[q]; part1 += d1 * d1; part2 += d2 * d2; }
                        }
                    }
                    const float bts = kTsBt[s];
                    float cs[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) cs[i] = kFwdShift[s][i];
#pragma unroll
                    for (int q = 0; q < NKD; ++q) {
                        if (q < nksD) R[L.k(s + 1) + fo + q * 64] = kv[q];
                        E[q] += bts * kv[q];
#pragma unroll
                        for (int i = 0; i < 5; ++i) Sa[i][q] = Sa[i + 1][q] + cs[i] * kv[q];
                    }
                }
                // embedded error estimate, SURVEY.md B.3
#pragma unroll
                for (int q = 0; q < NKD; ++q) {
                    if (q < nksD && colok && fb + fs * q < P.D) {
                        const float ut = dt * E[q];
                        const float sk = P.abstol + fmaxf(fabsf(up[q]), fabsf(un[q])) * P.reltol;
                        const float r = ut / sk;
                        part += r * r;
                    }
                }
            }
            CHAIN_STAMP(10);
            part = wave_sum_f(part); part1 = wave_sum_f(part1); part2 = wave_sum_f(part2);
            if (lane == 0) { RED[wave] = part; RED[kCW + wave] = part1; RED[2 * kCW + wave] = part2; }
            __syncthreads();
            if (tid == 0) {
                float s = 0.f, s1 = 0.f, s2 = 0.f;
                for (int w = 0; w < kCW; ++w) { s += RED[w]; s1 += RED[kCW + w]; s2 += RED[2 * kCW + w]; }
                float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
                ep[blockIdx.x] = s; ep[P.nwg + blockIdx.x] = s1; ep[2 * P.nwg + blockIdx.x] = s2;
            }
        }
    }
}

// fragment order <-> caller layout (debug entry points, and k1 hand-over of rnde_debug_attempt)
static __global__ void rnde_chain_convert_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int B, int ntiles, int nksD, int to_caller) {
    const long long total = (long long)ntiles * nksD * 64;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int lane = (int)(e & 63), q = (int)((e >> 6) % nksD), tile = (int)((e >> 6) / nksD);
        const int f = 4 * q + (lane >> 4), gcol = tile * 16 + (lane & 15);
        const bool ok = f < D && gcol < B;
        if (to_caller) { if (ok) dst[(size_t)gcol * D + f] = src[e]; }
        else dst[e] = ok ? src[(size_t)gcol * D + f] : 0.f;
    }
}

}  // namespace rnde
