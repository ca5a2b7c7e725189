// rnde_quad.h -- layout 1 of the chain engine: one wave owns FOUR batch columns (rnde_chain.h's layout 0 owns sixteen).
//
// Why: the chain engine is latency bound by construction (B = 512 is 32 waves of 16 columns on a 1024-SIMD chip, each
// running ~184 dependent MFMAs and ~70 tanh registers per f evaluation).  With 4 columns per wave the same work is
// spread over 128 waves, and a Dense layer becomes v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4 rows x 4 columns per
// instruction, i.e. up to 64 output rows at once, the reduction index k advancing one per instruction:
//   * A operand of block b, lane i  = W[4b + i][k]           (fragment tables in LDS, 16 bytes = 4 k per lane)
//   * B operand of block b, lane j  = act[k][column j]       (the previous layer's activations, wave-private LDS,
//                                                             read as a 16-way broadcast, 16 bytes = 4 k per lane)
//   * D registers (4 per lane)      = out[4b + i'][column j]
// Narrow layers (fewer than 16 row blocks) split K over the idle blocks: block s*nb + b accumulates rows 4b..4b+3 over the
// k range of split s, and the splits are summed when the activations are written back (one LDS round trip that the
// activation write needs anyway).  Everything a wave touches in LDS except the read-only tables is private to it: no
// workgroup barriers inside an f evaluation.  Because the activation vector lives in LDS, every loop over k is a rolled
// loop with a run-time trip count: no per-shape code, no run-time dispatch cost (layout 0 needs compile-time shapes for
// speed, see rnde_chain.h ALT).
// State arrays (uprev, k_j, ...) are 4 registers per lane: rows 4b..4b+3 (b = lane >> 2 < ceil(D/4)) of column lane & 3.
#pragma once
#include "rnde_device.h"

namespace rnde {

constexpr int kQW = 4;            // waves per workgroup (4 columns each)
constexpr int kQRS = 68;          // row stride (floats) of a wave's activation image [4 columns][kQRS]: 64 rows + bank skew
constexpr int kQMaxL = 8;

struct QuadGeo {
    int n_layers, time_dep, pre_act, D;
    int in[kQMaxL], out[kQMaxL], act[kQMaxL], poff[kQMaxL];
    int nb[kQMaxL], ks[kQMaxL], kper[kQMaxL];          // row blocks, K splits, k per split (multiple of 4) -- forward
    int aoff[kQMaxL], boff[kQMaxL];                    // forward A table / bias table offsets (units of 64 f32x4 = 1 KiB)
    int nbT[kQMaxL], ksT[kQMaxL], kperT[kQMaxL], toff[kQMaxL];   // the same for the transposed products of the reverse pass
    int units_f, units_b, units_t;                     // table sizes in 1 KiB units: forward A | bias (+ time column) | transposed A
};

__host__ inline void quad_split(int rows, int kdim, int& nb, int& ks, int& kper) {
    nb = (rows + 3) / 4;
    ks = 16 / nb; if (ks > 4) ks = 4; if (ks < 1) ks = 1;
    kper = (((kdim + ks - 1) / ks) + 3) / 4 * 4;
}

// fragment tables: forward A (layer l, step group t4): lane (blk, i) holds W_l[4b + i][s*kper + 4*t4 + q], q = 0..3, where
// s = blk / nb, b = blk % nb; transposed tables the same with W_l^T; bias table: lane (blk, .) holds b_l[4 blk + q]
// (and, when time_dep, a second entry with the time column W_l[:, in]).
__global__ void rnde_quad_pack_kernel(const float* __restrict__ p, f32x4* __restrict__ tab, const QuadGeo G) {
    const long long total = (long long)(G.units_f + G.units_b + G.units_t) * 64;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        int unit = (int)(e >> 6);
        const int lane = (int)(e & 63), blk = lane >> 2, i = lane & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (unit < G.units_f) {
            int l = 0;
            while (l + 1 < G.n_layers && unit >= G.aoff[l + 1]) ++l;
            const int t4 = unit - G.aoff[l], s = blk / G.nb[l], b = blk - s * G.nb[l], row = 4 * b + i;
            if (s < G.ks[l] && row < G.out[l]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kl = 4 * t4 + q, k = s * G.kper[l] + kl;
                    if (kl < G.kper[l] && k < G.in[l]) v[q] = p[G.poff[l] + (size_t)k * G.out[l] + row];
                }
            }
        } else if (unit < G.units_f + G.units_b) {
            unit -= G.units_f;
            int l = 0;
            while (l + 1 < G.n_layers && unit >= G.boff[l + 1]) ++l;
            const bool tcol = (unit - G.boff[l]) == 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * blk + q;
                if (row < G.out[l]) v[q] = tcol ? p[G.poff[l] + (size_t)G.in[l] * G.out[l] + row] : p[G.poff[l] + (size_t)(G.in[l] + G.time_dep) * G.out[l] + row];
            }
        } else {
            unit -= G.units_f + G.units_b;
            int l = 0;
            while (l + 1 < G.n_layers && unit >= G.toff[l + 1]) ++l;
            const int t4 = unit - G.toff[l], s = blk / G.nbT[l], b = blk - s * G.nbT[l], row = 4 * b + i;   // row of W^T = input feature
            if (s < G.ksT[l] && row < G.in[l]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kl = 4 * t4 + q, k = s * G.kperT[l] + kl;                                     // k = output feature
                    if (kl < G.kperT[l] && k < G.out[l]) v[q] = p[G.poff[l] + (size_t)row * G.out[l] + k];
                }
            }
        }
        tab[e] = v;
    }
}

// ---- layer plan -------------------------------------------------------------------------------------------------------
// Everything a lane needs to know about a layer, precomputed once per launch into LDS (the first version re-derived it per
// layer per evaluation: a software integer division and half a dozen dependent scalar loads of QuadGeo per layer cost more
// than the layer's arithmetic).  Entry 0 of a layer is uniform, entries 1..64 are per lane.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kQPlanStride = 65;                       // i32x4 entries per layer
__device__ __forceinline__ void quad_build_plan(const QuadGeo& G, i32x4* plan, bool transposed, int tid, int nthreads) {
    for (int idx = tid; idx < G.n_layers * kQPlanStride; idx += nthreads) {
        const int l = idx / kQPlanStride, e = idx - l * kQPlanStride;
        const int nb = transposed ? G.nbT[l] : G.nb[l], ks = transposed ? G.ksT[l] : G.ks[l], kper = transposed ? G.kperT[l] : G.kper[l];
        i32x4 v;
        if (e == 0) v = (i32x4){kper >> 2, transposed ? G.toff[l] : G.aoff[l], G.boff[l], ks | (G.act[l] << 8) | (nb << 16)};
        else {
            const int lane = e - 1, blk = lane >> 2, j = lane & 3, s = blk / nb, b = blk - s * nb;
            v = (i32x4){j * kQRS + s * kper,                                                  // this lane's first activation row (B operand)
                        (s > 0 && s < ks) ? ((s - 1) * 4 + j) * kQRS + 4 * b : -1,              // where its K-split partial goes
                        blk < nb ? 1 : 0,                                                       // it owns output rows 4 blk .. 4 blk + 3
                        j * kQRS + 4 * blk};                                                    // where those go in the next activation image
        }
        plan[idx] = v;
    }
}

// D = A(table) * act_in over K, for all 16 blocks: 4 independent accumulation chains (a dependent 4x4x1 MFMA issues every
// ~40 cycles, an independent one every 8); the operands of the next TWO step groups are in flight while the MFMAs issue.
__device__ __forceinline__ f32x4 quad_mm(const f32x4* __restrict__ tab /* + lane */, const float* __restrict__ bsrc, int n4) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    f32x4 a0 = tab[0], b0 = *(const f32x4*)bsrc, a1 = a0, b1 = b0;
    if (n4 > 1) { a1 = tab[64]; b1 = *(const f32x4*)(bsrc + 4); }
    for (int t4 = 0; t4 < n4; ++t4) {
        f32x4 a2 = a1, b2 = b1;
        if (t4 + 2 < n4) { a2 = tab[(t4 + 2) * 64]; b2 = *(const f32x4*)(bsrc + 4 * (t4 + 2)); }
        c0 = mfma4(a0[0], b0[0], c0);
        c1 = mfma4(a0[1], b0[1], c1);
        c2 = mfma4(a0[2], b0[2], c2);
        c3 = mfma4(a0[3], b0[3], c3);
        a0 = a1; b0 = b1; a1 = a2; b1 = b2;
    }
    return (c0 + c1) + (c2 + c3);
}

__device__ __forceinline__ f32x4 quad_tanh4(const f32x4& v) {
    const f32x2 t01 = tanh_fast2((f32x2){v[0], v[1]}), t23 = tanh_fast2((f32x2){v[2], v[3]});
    return (f32x4){t01.x, t01.y, t23.x, t23.y};
}

// one product D = Table_l * act_in with the K splits summed: returns the 4 output rows of the lanes that own some (pl[2]).
__device__ __forceinline__ f32x4 quad_layer_mm(const i32x4& pu, const i32x4& pl, const f32x4* TAB, const float* ain, float* PT, int lane) {
    const int n4 = __builtin_amdgcn_readfirstlane(pu[0]), ks = __builtin_amdgcn_readfirstlane(pu[3]) & 255;
    f32x4 d = quad_mm(TAB + (size_t)__builtin_amdgcn_readfirstlane(pu[1]) * 64 + lane, ain + pl[0], n4);
    if (ks > 1) {
        if (pl[1] >= 0) *(f32x4*)(PT + pl[1]) = d;
        if (pl[2]) {
            const float* pr = PT + pl[3];
            for (int q = 1; q < ks; ++q) d += *(const f32x4*)(pr + (q - 1) * 4 * kQRS);
        }
    }
    return d;
}

// k = f(g, ts) for the wave's 4 columns.  g, out: 4 registers per lane (rows 4 blk + q of column j; lanes blk >= ceil(D/4): zeros).
// PLAN: forward layer plan; TAB: read-only tables (forward A | bias); AQ, BQ: wave-private activation images; PT: K-split partials.
// HS (reverse pass only, else nullptr): wave-private copies of every layer's input, [n_layers + 1][4][kQRS].
__device__ __forceinline__ void quad_eval(const QuadGeo& G, const i32x4* PLAN, const f32x4* TAB, float* AQ, float* BQ, float* PT, float ts,
                                          const f32x4& g, f32x4& out, int lane, float* HS = nullptr, int n_run = -1) {
    const int blk = lane >> 2, j = lane & 3;
    const f32x4* BT = TAB + (size_t)G.units_f * 64;
    const bool mine0 = blk < ((G.D + 3) >> 2);
    {
        const f32x4 a0 = G.pre_act ? quad_tanh4(g) : g;
        if (mine0) { *(f32x4*)(AQ + j * kQRS + 4 * blk) = a0; if (HS) *(f32x4*)(HS + j * kQRS + 4 * blk) = a0; }
    }
    float* ain = AQ; float* aout = BQ;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int nl = n_run < 0 ? G.n_layers : n_run;
#pragma unroll 1
    for (int l = 0; l < nl; ++l) {
        const i32x4 pu = PLAN[l * kQPlanStride], pl = PLAN[l * kQPlanStride + 1 + lane];
        const int boff = __builtin_amdgcn_readfirstlane(pu[2]);
        f32x4 bias = BT[(size_t)boff * 64 + lane];
        if (G.time_dep) bias += ts * BT[(size_t)(boff + 1) * 64 + lane];
        const f32x4 d = quad_layer_mm(pu, pl, TAB, ain, PT, lane);
        if (pl[2]) {
            v = d + bias;
            if ((__builtin_amdgcn_readfirstlane(pu[3]) >> 8) & 255) v = quad_tanh4(v);
            *(f32x4*)(aout + pl[3]) = v;
            if (HS) *(f32x4*)(HS + (size_t)(l + 1) * 4 * kQRS + pl[3]) = v;
        }
        float* t = ain; ain = aout; aout = t;
    }
    out = mine0 ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}

}  // namespace rnde
