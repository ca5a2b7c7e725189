// rnde_chainmw.h -- "multi-wave" kernels of the chain engine: the adaptive Tsit5 attempt for small-width Dense chains
// (every width <= 64; the latent-ODE dynamics of reference experiments/latent_ode.jl:113-124) with
// ONE WORKGROUP OF FOUR WAVES PER 16 BATCH COLUMNS instead of one wave (rnde_chain.h).
//
// Why: with one wave per tile a Dense layer is a chain of 20-26 dependent-ish MFMAs plus 13 tanh registers on a single SIMD:
// ~2.2 k cycles per layer, 48 layers per attempted step, 32 waves on a 1,024-SIMD chip (the kernel furthest
// below its roof).  Here a layer's output tiles are dealt to the four SIMDs of a CU:
//   * activations live in LDS as [feature][16 columns] (two ping-pong buffers); an MFMA's B operand for k-step
//     (input tile mi, j) is one conflict-free ds_read_b32 at X + (16 mi + 4 j) * 16 + lane;
//   * wave w owns output tile w of every layer: <= 16 MFMAs (k-steps padded to whole input tiles, zero weights), its 4 D
//     registers get bias / tanh and go back to LDS at Y[(16 mo + 4 g + i) * 16 + col]; one workgroup barrier per layer;
//   * weight fragments stay resident in LDS (natural row order: no K permutation is needed when operands come from LDS);
//   * the Runge-Kutta state (uprev, running stage sums, error accumulator) is element-wise work: 256 lanes x NKD/4 registers
//     per array, element e = tid + 256 r <-> (feature e >> 4, column e & 15) -- the arena's fragment order IS [feature][16],
//     so tape records, initialisation buffers and the host loop are those of rnde_chain.h, unchanged.
// The forward pass tapes every layer's input (the slab rows the parameter-gradient kernel reads anyway), so the reverse pass
// (rnde_bchainmw.h) recomputes nothing: 8 layer products per evaluation instead of 16.
#pragma once
#include "rnde_chain.h"

namespace rnde {

constexpr int kMwWaves = 4;
constexpr int kMwThreads = 64 * kMwWaves;

struct MwGeo {
    int n_layers, time_dep, pre_act, D;
    int width[kCMaxL + 1], mt[kCMaxL + 1];   // mt = 16-feature tiles of each width (k-steps are padded to 4 mt)
    int act[kCMaxL], poff[kCMaxL];
    int foff[kCMaxL], toff[kCMaxL];          // fragment offsets (units of 64 floats) inside the forward / transposed tables
    int nfrag_f, nfrag_t;                    // padded to multiples of 4 (1 KiB DMA units)
    int hrow[kCMaxL + 1], zrow[kCMaxL], RS;  // slab rows (k-step units) of layer inputs / pre-activation cotangents, as BChainParams
};
// global table: [forward fragments | bias vectors 8 x 64 | time columns 8 x 64 | transposed fragments]
__host__ __device__ inline size_t mw_tab_floats(const MwGeo& G) { return (size_t)(G.nfrag_f + G.nfrag_t) * 64 + 1024; }

// An explicit Runge-Kutta pair as DATA, in first-same-as-last form: S stages of which the last one evaluates f at u_new (row S - 1 of the
// stage matrix = the weights of u_new, c = 1), embedded error weights, the order the step-size controller uses.  Kernels instantiated with
//   TAB = 1  take a 7-stage pair with a quartic dense output from here instead of the Tsit5 constants of rnde_device.h (RNDE_SOLVER_DP5:
//            Dormand-Prince 5(4), validated against scipy's RK45),
//   TAB = 2  take ANY pair of up to kRkSMax stages: stage loops, tape records and evaluation counts run on S (RNDE_SOLVER_DOP853: scipy's
//            12 stages + the closing evaluation = 13, csrc/rk_tables.h; no dense output, no stiffness estimate).  A pair that is not
//            first-same-as-last (Verner's Vern7: 10 stages) is the same table with one more row whose error weight is 0.
constexpr int kRkSMax = 13;
struct RkTab {
    int S, order;                        // 7 / 5 unless TAB = 2
    float fwd[kRkSMax][kRkSMax - 1];     // fwd[s][i] = a_{s+1+i, s}   (as kFwdShift; 0 beyond the tableau)
    float bwd[kRkSMax][kRkSMax - 1];     // bwd[s][i] = a_{s, s-1-i}   (as kBwdShift)
    float aN[kRkSMax + 1];               // a_{S-1, j}: the weights of u_new
    float bt[kRkSMax + 1], c[kRkSMax + 1];   // embedded error weights, nodes
    float dense[7][4];                   // 7-stage pairs: b_i(theta) = sum_j dense[i][j] theta^(j+1)
};
template <int TAB> __device__ __forceinline__ float rk_fwd(const RkTab& T, int s, int i) { return TAB ? T.fwd[s][i] : kFwdShift[s][i]; }
template <int TAB> __device__ __forceinline__ float rk_bwd(const RkTab& T, int s, int i) { return TAB ? T.bwd[s][i] : kBwdShift[s][i]; }
template <int TAB> __device__ __forceinline__ float rk_bt(const RkTab& T, int j) { return TAB ? T.bt[j] : kTsBt[j]; }
template <int TAB> __device__ __forceinline__ float rk_c(const RkTab& T, int s) { return TAB ? T.c[s] : kTsC[s]; }
template <int TAB> __device__ __forceinline__ float rk_a7(const RkTab& T, int j) { return TAB ? T.aN[j] : kTsA[6][j]; }
template <int TAB> __device__ __forceinline__ void rk_dense(const RkTab& T, float th, float (&b)[7]) {
    if (TAB) {
#pragma unroll
        for (int i = 0; i < 7; ++i) b[i] = th * (T.dense[i][0] + th * (T.dense[i][1] + th * (T.dense[i][2] + th * T.dense[i][3])));
    } else dense_weights(th, b);
}
template <int TAB> __device__ __forceinline__ void rk_dense_deriv(const RkTab& T, float th, float (&db)[7]) {
    if (TAB) {
#pragma unroll
        for (int i = 0; i < 7; ++i) db[i] = T.dense[i][0] + th * (2.f * T.dense[i][1] + th * (3.f * T.dense[i][2] + th * 4.f * T.dense[i][3]));
    } else dense_weights_deriv(th, db);
}

struct MwParams {
    StepParams F;
    MwGeo G;
    RkTab rk;
    const float* tab;
    float* u_out;            // MW_FINISH: end state, D x B caller layout (may be NULL)
    float* slab;             // [n_evals][ntiles][RS][64] (NULL when not taping): evaluation 0 = f(u0), 1 = f(u1), 2 + 6 n + (s - 1) = stage s of attempt n
    long long ev_stride;
    int ntiles;
    // MW_SOLVE (the whole adaptive solve in one launch): the workgroups meet once per attempted step -- through the L2 of ONE XCD while they
    // fit it (<= 32 tiles), through agent-scope entries on the memory side otherwise (xch_global: <= 256 tiles, every workgroup resident)
    unsigned long long* xch;     // [max_attempts][3][ntiles] {float value, uint tag}: every entry written once per solve
    unsigned* xcc;               // [ntiles] HW_REG_XCC_ID of each workgroup (the host checks they agree; not looked at when xch_global)
    unsigned* abort_word;        // a meeting timed out
    unsigned epoch;              // tag = epoch * 8192 + attempt + 1
    int n_limit;                 // attempts this launch may run (the activation slab is sized for that many): reaching it ends the launch with done = 0
    int xch_global;              // 0: 8 x ntiles workgroups launched, those with blockIdx % 8 == xcd_slot work (one XCD); 1: ntiles launched, all of them work
    int xcd_slot;                // which residue of blockIdx % 8 works (per handle: two handles on two streams then meet on different XCDs)
};

// cross-workgroup sums of the three norm partials of attempt `seq` (MW_SOLVE): every workgroup publishes {value, tag} and polls the others'.
// One XCD (global = 0): a plain 8-byte store (written through to the XCD's L2) and L1-bypassing loads -- the SDE engine's meeting (rnde_sde.h
// sde_exchange).  Any placement (global = 1, up to 256 workgroups): agent-scope stores and loads, the meeting of rnde_stage_solve.h.
// Called by wave 0; `mine` valid in lane 0; the sums come out in the order sum_partials forms them (lane l adds entries l, l + 64, l + 128,
// l + 192 in double, then the wave reduction), so a solve is bit-identical to the one-launch-per-attempt path.  false: timed out / aborted.
constexpr int kMwSpinMax = 4000000;
constexpr int kMwMeetMax = 256;      // workgroups one meeting can hold (= CUs: every participant must be resident)
struct MwMeet { unsigned long long* xch; unsigned* abort_word; unsigned epoch; int ntiles; int global; };     // what a meeting needs (forward solve, reverse sweep)
__device__ __forceinline__ bool mw_exchange3(const MwMeet& Q, int seq, const float (&mine)[3], double (&out)[3], int tile, int lane) {
    const unsigned tag = Q.epoch * 8192u + (unsigned)seq + 1u;
    unsigned long long* base = Q.xch + (size_t)seq * 3 * Q.ntiles;
    if (lane == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float m = mine[v];
            if (m != m) m = __uint_as_float(0x7FC00000u);
            const unsigned long long e = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(m);
            if (Q.global) __hip_atomic_store(base + (size_t)v * Q.ntiles + tile, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else base[(size_t)v * Q.ntiles + tile] = e;
        }
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    typedef unsigned mw_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        double sv = 0.0;
        for (int b0 = 0; b0 < Q.ntiles; b0 += 64) {
            const int i = b0 + lane;
            unsigned long long e = 0;
            bool ok = i >= Q.ntiles;
            int spins = 0;
            while (true) {
                if (!ok) {
                    if (Q.global) e = __hip_atomic_load(base + (size_t)v * Q.ntiles + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else {
                        __asm__ volatile("" ::: "memory");
                        const mw_u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(((size_t)v * Q.ntiles + i) * 8), 0, 16);   // aux 16 = sc1: misses L1
                        e = ((unsigned long long)q.y << 32) | q.x;
                    }
                    ok = (unsigned)(e >> 32) == tag;
                }
                if (__all(ok)) break;
                if (++spins > kMwSpinMax || ((spins & 1023) == 0 && __hip_atomic_load(Q.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (lane == 0) __hip_atomic_store(Q.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
            if (i < Q.ntiles) sv += (double)__uint_as_float((unsigned)(e & 0xFFFFFFFFull));
        }
        out[v] = wave_sum_d(sv);
    }
    return true;
}

// forward fragment (l, mo, mi, j): lane (kk = lane >> 4, rho = lane & 15) = W_l[16 mo + rho][16 mi + 4 j + kk]
// transposed      (l, mi, mo, j): lane (kk, rho)                         = W_l[16 mo + 4 j + kk][16 mi + rho]
// bias vector l: b_l[f] for f < 64; time column l: W_l[f][in] (TDChain layers)
static __global__ void rnde_chainmw_pack_kernel(const float* __restrict__ p, float* __restrict__ tab, const MwGeo G) {
    const long long nf = (long long)G.nfrag_f * 64, nt = (long long)G.nfrag_t * 64, total = nf + 1024 + nt;
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        float v = 0.f;
        if (e < nf || e >= nf + 1024) {
            const bool tr = e >= nf;
            const long long ee = tr ? e - nf - 1024 : e;
            int fr = (int)(ee >> 6);
            const int lane = (int)(ee & 63), rho = lane & 15, kk = lane >> 4;
            const int* off = tr ? G.toff : G.foff;
            int l = 0;
            while (l + 1 < G.n_layers && fr >= off[l + 1]) ++l;
            fr -= off[l];
            const int in = G.width[l], out = G.width[l + 1];
            const int j = fr & 3, t = fr >> 2;
            int fo, fi;
            if (!tr) { const int mo = t / G.mt[l], mi = t - mo * G.mt[l]; fo = 16 * mo + rho; fi = 16 * mi + 4 * j + kk; }
            else { const int mi = t / G.mt[l + 1], mo = t - mi * G.mt[l + 1]; fo = 16 * mo + 4 * j + kk; fi = 16 * mi + rho; }
            if (t < G.mt[l] * G.mt[l + 1] && fo < out && fi < in) v = p[G.poff[l] + (size_t)fi * out + fo];
        } else {
            const int r = (int)(e - nf), tcol = r >= 512, rr = r & 511, l = rr >> 6, f = rr & 63;
            if (l < G.n_layers && f < G.width[l + 1]) {
                const int in = G.width[l], out = G.width[l + 1];
                v = tcol ? (G.time_dep ? p[G.poff[l] + (size_t)in * out + f] : 0.f) : p[G.poff[l] + (size_t)(in + G.time_dep) * out + f];
            }
        }
        tab[e] = v;
    }
}

__device__ __forceinline__ void mw_fill_lds(const float* __restrict__ src, float* dst, int units, int wave, int lane) {
    for (int u = wave; u < units; u += kMwWaves) dma_unit((const f32x4*)(src + (size_t)u * 256) + lane, dst + (size_t)u * 256);
    wait_vm<0>();
    __syncthreads();
}

// One Dense layer: X (LDS, [feature][16], rows < 16 mt[l] finite) -> Y.  Wave w computes output tile w.  Ends with a barrier.
// hs != NULL: the layer's OUTPUT is also written to the slab rows hs (the next layer's input / the evaluation's value).
__device__ __forceinline__ void mw_layer(const MwGeo& G, const float* FRm, const float* BV, const float* TV, int l, float ts, const float* X, float* Y,
                                         float* __restrict__ hs, int wave, int lane) {
    const int mtin = G.mt[l], mtout = G.mt[l + 1];
    if (wave < mtout) {
        const int mo = wave, g = lane >> 4, col = lane & 15;
        f32x4 acc0 = *(const f32x4*)(BV + l * 64 + 16 * mo + 4 * g), acc1 = {0.f, 0.f, 0.f, 0.f};
        if (G.time_dep) { const f32x4 wt = *(const f32x4*)(TV + l * 64 + 16 * mo + 4 * g); acc0 = __builtin_elementwise_fma((f32x4){ts, ts, ts, ts}, wt, acc0); }
        const float* fr = FRm + ((size_t)G.foff[l] + (size_t)mo * mtin * 4) * 64 + lane;
        const float* xb = X + lane;
        float a[4], b[4], a2[4], b2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = fr[j * 64]; b[j] = xb[j * 64]; }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (mi < mtin) {
                if (mi + 1 < mtin) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a2[j] = fr[((mi + 1) * 4 + j) * 64]; b2[j] = xb[((mi + 1) * 4 + j) * 64]; }
                }
                acc0 = mfma16(a[0], b[0], acc0); acc1 = mfma16(a[1], b[1], acc1);
                acc0 = mfma16(a[2], b[2], acc0); acc1 = mfma16(a[3], b[3], acc1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[j] = a2[j]; b[j] = b2[j]; }
            }
        }
        f32x4 o = acc0 + acc1;
        if (G.act[l] != 0) {
            const f32x2 t01 = tanh_fast2((f32x2){o[0], o[1]}), t23 = tanh_fast2((f32x2){o[2], o[3]});
            o = (f32x4){t01.x, t01.y, t23.x, t23.y};
        }
        float* yp = Y + (16 * mo + 4 * g) * 16 + col;
#pragma unroll
        for (int i = 0; i < 4; ++i) yp[i * 16] = o[i];
        if (hs) {
            float* sp = hs + (16 * mo + 4 * g) * 16 + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) sp[i * 16] = o[i];
        }
    }
    __syncthreads();
}

#ifdef RNDE_DIAG
#define MW_STAMP(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[i] = clock64(); } while (0)
#else
#define MW_STAMP(i) do { } while (0)
#endif

// k = f(g, ts) for the workgroup's 16 columns.  gv / kv: element-wise registers (e = tid + 256 r).  XB, YB: 64 x 16 floats each.
// sl: slab base of this (evaluation, tile) or NULL.
template <int NR>
__device__ __forceinline__ void mw_eval(const MwGeo& G, const float* FRm, const float* BV, const float* TV, float* XB, float* YB, float ts,
                                        const float (&gv)[NR], float (&kv)[NR], float* __restrict__ sl, int tid, int wave, int lane,
                                        unsigned long long* dbg = nullptr) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float a = G.pre_act ? tanh_fast(gv[r]) : gv[r];
        XB[tid + 256 * r] = a;
        if (sl) sl[(size_t)G.hrow[0] * 64 + tid + 256 * r] = a;
    }
    // (a layer reads exactly the 16 mt[l] rows its predecessor wrote; the input covers 4 NKD >= 16 mt[0] rows: nothing stale is ever read)
    __syncthreads();
    // (rolled on purpose: eight unrolled copies of the layer body measured 41.5 us per attempt against 38.6 us)
    float* X = XB; float* Y = YB;
#pragma unroll 1
    for (int l = 0; l < G.n_layers; ++l) {
        MW_STAMP(16 + l);
        mw_layer(G, FRm, BV, TV, l, ts, X, Y, sl ? sl + (size_t)G.hrow[l + 1] * 64 : nullptr, wave, lane);
        float* t_ = X; X = Y; Y = t_;
    }
    MW_STAMP(16 + G.n_layers);
#pragma unroll
    for (int r = 0; r < NR; ++r) kv[r] = ((tid + 256 * r) >> 4) < 16 * G.mt[G.n_layers] ? X[tid + 256 * r] : 0.f;   // (rows past the last tile were never written)
    __syncthreads();   // X is rewritten by the next evaluation's input
}

// ---- the reference's latent-ODE dynamics (experiments/latent_ode.jl:113-124: 20 -> 50 -> 20 -> ... eight Dense layers, time independent) with
// ---- the weights REGISTER STATIONARY: LAT = 1 instantiations.  mw_layer reads a layer's A fragments and its geometry every time it
// ---- runs (16 + 16 LDS reads and a handful of scalar loads in front of at most 16 MFMAs, 48 times per attempted step); here a wave
// ---- loads the fragments of ITS output tile of every layer once per launch (8 x 8 or 16 registers), the layer loop is unrolled with the
// ---- tile counts as constants, and the fragment table never goes to LDS at all.  Same arithmetic in the same order as mw_layer.
constexpr int kLatLayers = 8;
__host__ __device__ constexpr int lat_mt(int i) { return (i & 1) ? 4 : 2; }     // 16-feature tiles of width i: 20 (2 tiles), 50 (4 tiles), 20, ...
__host__ __device__ constexpr int lat_width(int i) { return (i & 1) ? 50 : 20; }
// k-steps of 4 that hold features of width i: 5 of the 8 a 2-tile operand has, 13 of 16.  The steps behind them multiply the zero padding of the
// fragments -- a sum that gains exact zeros -- and are not issued: 72 MFMAs per evaluation instead of 96, same values
__host__ __device__ constexpr int lat_ks(int i) { return (lat_width(i) + 3) / 4; }
struct LatWeights { float a[kLatLayers][16]; f32x4 b[kLatLayers]; };
__device__ __forceinline__ void lat_load(const MwGeo& G, const float* __restrict__ tab, LatWeights& W, int wave, int lane) {
    const float* BVg = tab + (size_t)G.nfrag_f * 64;
    const int g = lane >> 4;
#pragma unroll
    for (int l = 0; l < kLatLayers; ++l) {
        constexpr int dummy = 0; (void)dummy;
        const int mtin = lat_mt(l), mtout = lat_mt(l + 1);
        const int mo = wave < mtout ? wave : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) W.a[l][j] = (j < lat_ks(l)) ? tab[((size_t)G.foff[l] + (size_t)mo * mtin * 4 + (j < lat_ks(l) ? j : 0)) * 64 + lane] : 0.f;
        W.b[l] = *(const f32x4*)(BVg + l * 64 + 16 * mo + 4 * g);
    }
}
template <int NR>
__device__ __forceinline__ void mw_eval_lat(const MwGeo& G, const LatWeights& W, float* XB, float* YB, const float (&gv)[NR], float (&kv)[NR],
                                            float* __restrict__ sl, int tid, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float a = G.pre_act ? tanh_fast(gv[r]) : gv[r];
        XB[tid + 256 * r] = a;
        if (sl) sl[(size_t)G.hrow[0] * 64 + tid + 256 * r] = a;
    }
    __syncthreads();
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int l = 0; l < kLatLayers; ++l) {
        float* X = (l & 1) ? YB : XB;
        float* Y = (l & 1) ? XB : YB;
        constexpr int kDummy = 0; (void)kDummy;
        const int mtin = lat_mt(l), mtout = lat_mt(l + 1);
        if (wave < mtout) {
            const float* xb = X + lane;
            float b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j < lat_ks(l)) b[j] = xb[j * 64];
            f32x4 acc0 = W.b[l], acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) {      // even k-steps into acc0, odd ones into acc1, ascending (as before)
                if (j < lat_ks(l)) { if (j & 1) acc1 = mfma16(W.a[l][j], b[j], acc1); else acc0 = mfma16(W.a[l][j], b[j], acc0); }
            }
            f32x4 o = acc0 + acc1;
            if (G.act[l] != 0) {
                const f32x2 t01 = tanh_fast2((f32x2){o[0], o[1]}), t23 = tanh_fast2((f32x2){o[2], o[3]});
                o = (f32x4){t01.x, t01.y, t23.x, t23.y};
            }
            float* yp = Y + (16 * wave + 4 * g) * 16 + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) yp[i * 16] = o[i];
            if (sl) {
                float* sp = sl + (size_t)G.hrow[l + 1] * 64 + (16 * wave + 4 * g) * 16 + col;
#pragma unroll
                for (int i = 0; i < 4; ++i) sp[i * 16] = o[i];
            }
        }
        __syncthreads();
    }
    float* X = (kLatLayers & 1) ? YB : XB;
#pragma unroll
    for (int r = 0; r < NR; ++r) kv[r] = ((tid + 256 * r) >> 4) < 16 * lat_mt(kLatLayers) ? X[tid + 256 * r] : 0.f;
    __syncthreads();
}

enum { MW_STEP = 0, MW_INIT_A = 1, MW_INIT_B = 2, MW_FEVAL = 3, MW_FINISH = 4, MW_SOLVE = 5 };

// NR = NKD / 4 registers per state array and lane (NKD = 4, 8, 16 k-steps of D as in rnde_chain.h: arena arrays are NKD * 64 floats per tile)
template <int NR, int MODE, int TAB = 0, int LAT = 0>
__global__ __launch_bounds__(kMwThreads) void rnde_chainmw_kernel(const MwParams Q, const int n) {
    const StepParams& P = Q.F;
    const MwGeo& G = Q.G;
    constexpr int NKD = 4 * NR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* FRm = smem;
    float* BV = FRm + (size_t)G.nfrag_f * 64;
    float* TV = BV + 512;
    float* XB = TV + 512;
    float* YB = XB + 1024;
    float* RED = YB + 1024;    // [3][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // MW_SOLVE: 8 x ntiles workgroups are launched and those with blockIdx % 8 != 0 leave at once, so that the ones that work share ONE XCD
    // (round-robin dispatch; every workgroup records its XCC id, the host checks they agree): they meet through that L2 once per attempt
    // (xch_global, more than 32 tiles: every launched workgroup works and the meeting goes through the memory side)
    if constexpr (MODE == MW_SOLVE) { if (!Q.xch_global && (int)(blockIdx.x & 7) != Q.xcd_slot) return; }
    const int tile = (MODE == MW_SOLVE && !Q.xch_global) ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const bool writer = (tile == 0 && tid == 0);
    if constexpr (MODE == MW_SOLVE) { if (tid == 0) Q.xcc[tile] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
    constexpr int SM = TAB == 2 ? kRkSMax : 7;               // stages the register arrays are sized for
    const int NS = TAB == 2 ? Q.rk.S : 7;                    // stages of the pair (first-same-as-last form)
    const ChainRec L{(long long)Q.ntiles * NKD * 64, NS};
    const size_t fo = (size_t)tile * NKD * 64 + tid;         // element r of this lane: fo + 256 r
    unsigned long long* dbg = nullptr;
#ifdef RNDE_DIAG
    if (MODE == MW_STEP) dbg = (unsigned long long*)P.dbg_out;
#endif
    MW_STAMP(0);
    // (STEP) what the controller needs of attempt n - 1 -- its state, its error partials -- and the record it wrote (the likely starting state)
    // are requested in front of the weight loads, as in rnde_stage_attempt_kernel: a wave's loads return in order, and behind the weights these
    // three were three serial cold round trips in front of every attempt of a solve
    float pre_part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 prev_raw[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float sp_up[NR], sp_k[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { sp_up[r] = 0.f; sp_k[r] = 0.f; }
    const bool pre = MODE == MW_STEP && n > 0;
    const bool spec = pre && P.tape && !P.forced;
    if constexpr (MODE == MW_STEP) {
        if (pre) {
            const f32x4* cp = (const f32x4*)&P.ctl[(n - 1) & 1];
            prev_raw[0] = cp[0]; prev_raw[1] = cp[1]; prev_raw[2] = cp[2];
            partials_request(P.errpart + (size_t)((n - 1) & 1) * 3 * P.nwg, lane, pre_part);
        }
        if (spec) {
            const float* Rs = P.arena + (long long)(n - 1) * P.rec_stride;
#pragma unroll
            for (int r = 0; r < NR; ++r) { sp_up[r] = Rs[L.unew() + fo + 256 * r]; sp_k[r] = Rs[L.k(NS) + fo + 256 * r]; }
        }
    }
    // LAT: this wave's weight fragments in registers for the whole launch, nothing to fill (the LDS layout is kept: XB / YB sit where they sit)
    LatWeights LW;
    if constexpr (MODE != MW_FINISH && LAT) lat_load(G, Q.tab, LW, wave, lane);
    if constexpr (MODE != MW_FINISH && !LAT) mw_fill_lds(Q.tab, smem, (G.nfrag_f >> 2) + 4, wave, lane);
    MW_STAMP(1);
    auto eval = [&](float ts, const float (&gin)[NR], float (&kout)[NR], float* slp, unsigned long long* dbgp) {
        if constexpr (LAT) mw_eval_lat<NR>(G, LW, XB, YB, gin, kout, slp, tid, wave, lane);
        else mw_eval<NR>(G, FRm, BV, TV, XB, YB, ts, gin, kout, slp, tid, wave, lane, dbgp);
    };
    // element (r): feature f = (tid + 256 r) >> 4, column gcol
    const int gcol = tile * 16 + (tid & 15);
    const bool colok = gcol < P.B;
    auto feat = [&](int r) { return (tid + 256 * r) >> 4; };
    auto valid = [&](int r) { return colok && feat(r) < P.D; };
    auto ldx = [&](const float* base, int r) { return valid(r) ? base[(size_t)gcol * P.D + feat(r)] : 0.f; };
    float* slab_tile = Q.slab ? Q.slab + ((size_t)tile * G.RS) * 64 : nullptr;

    if constexpr (MODE == MW_FEVAL) {
        float gv[NR], kv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) gv[r] = ldx(P.x, r);
        eval(P.forced_t, gv, kv, nullptr, nullptr);
#pragma unroll
        for (int r = 0; r < NR; ++r) if (valid(r)) P.dbg_out[(size_t)gcol * P.D + feat(r)] = kv[r];
        return;
    } else if constexpr (MODE == MW_INIT_A || MODE == MW_INIT_B) {
        // ---- initial-step heuristic, SURVEY.md B.1 (same arithmetic as rnde_chain_kernel) ----
        float dt0 = 0.f;
        if constexpr (MODE == MW_INIT_B) {
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
        float xv[NR], fv[NR], gv[NR], kv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            xv[r] = ldx(P.x, r);
            fv[r] = 0.f;
            if constexpr (MODE == MW_INIT_B) {
                fv[r] = P.f0[fo + 256 * r];
                gv[r] = xv[r] + dt0 * fv[r];
                P.u1[fo + 256 * r] = gv[r];
            } else gv[r] = xv[r];
        }
        float* sl = (slab_tile && P.tape) ? slab_tile + (size_t)(MODE == MW_INIT_B ? 1 : 0) * Q.ev_stride : nullptr;
        eval((MODE == MW_INIT_B) ? P.t0 + dt0 : P.t0, gv, kv, sl, nullptr);
        float pa = 0.f, pb = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ((MODE == MW_INIT_B) ? P.f1 : P.f0)[fo + 256 * r] = kv[r];
            if (valid(r)) {
                const float sk = P.abstol + fabsf(xv[r]) * P.reltol;
                if constexpr (MODE == MW_INIT_A) { const float a = xv[r] / sk, b = kv[r] / sk; pa += a * a; pb += b * b; }
                else { const float a = (kv[r] - fv[r]) / sk; pa += a * a; }
            }
        }
        pa = wave_sum_f(pa); pb = wave_sum_f(pb);
        if (lane == 0) { RED[wave] = pa; RED[4 + wave] = pb; }
        __syncthreads();
        if (tid == 0) {
            float sa = 0.f, sb = 0.f;
            for (int w = 0; w < kMwWaves; ++w) { sa += RED[w]; sb += RED[4 + w]; }
            if constexpr (MODE == MW_INIT_A) { P.initpart[tile] = sa; P.initpart[P.nwg + tile] = sb; }
            else P.initpart[2 * P.nwg + tile] = sa;
        }
        return;
    } else {
        // ---- controller, then (STEP) one attempted step / (FINISH) the last step's save points and the copy-out ----
        // (MW_SOLVE: the same, attempt after attempt, until the controller says done; `nn` is the attempt, the previous state and the
        //  norm sums of the previous attempt are carried in registers instead of being re-read behind a kernel boundary)
        StepState Sprev{};
        double xs[3] = {0.0, 0.0, 0.0};
        float* RED2 = RED + 16;                  // (MW_SOLVE: the three sums, as doubles, for the waves that did not meet)
      for (int nn = n; ; ++nn) {
        StepState S;
        if constexpr (MODE == MW_SOLVE) {
            const float none[4] = {0.f, 0.f, 0.f, 0.f};
            if (nn == 0) S = advance_state(P, 0, lane, writer, &P.ctl[0]);
            else S = advance_state_t<true>(P, nn, lane, writer, &P.ctl[nn & 1], none, Sprev, xs);
            if (!S.done && nn >= Q.n_limit) { if (writer) *P.ctl_final = S; return; }      // out of slab: the host regrows it and solves again
        } else if (pre) {
            asm volatile("" : "+v"(prev_raw[0]), "+v"(prev_raw[1]), "+v"(prev_raw[2]));   // (first use: keeps the unpacking, and its wait, down here)
            static_assert(sizeof(StepState) == 48, "StepState is read as three 16-byte words");
            StepState prev_state;
            __builtin_memcpy(&prev_state, prev_raw, sizeof(StepState));
            prev_state.live = __builtin_amdgcn_readfirstlane(prev_state.live); prev_state.done = __builtin_amdgcn_readfirstlane(prev_state.done);
            S = advance_state_t<true>(P, n, lane, writer, &P.ctl[n & 1], pre_part, prev_state);
        } else S = advance_state(P, n, lane, writer, (MODE == MW_FINISH) ? P.ctl_final : &P.ctl[n & 1]);
        MW_STAMP(2);
        if (P.nsave > 0) {
            // saveat ({R,true} methods, neural_ode.jl:79-108): the points inside the step accepted last
            if (nn == 0) {
                if (S.next_save > 0) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) if (valid(r)) P.sv_out[((size_t)gcol * P.nsave) * P.D + feat(r)] = P.x[(size_t)gcol * P.D + feat(r)];
                }
            } else {
                const StepState pv = MODE == MW_SOLVE ? Sprev : P.ctl[(nn - 1) & 1];
                const int lo = pv.next_save, hi = S.next_save;
                if (hi > lo && !pv.done) {
                    const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                    const float* Rp = P.arena + (long long)S.live * P.rec_stride;
                    float up[NR], un[NR], k[7][NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        up[r] = Rp[L.upc() + fo + 256 * r]; un[r] = Rp[L.unew() + fo + 256 * r]; k[0][r] = Rp[L.k1c() + fo + 256 * r];
#pragma unroll
                        for (int j = 1; j < 7; ++j) k[j][r] = Rp[L.k(j + 1) + fo + 256 * r];
                    }
                    for (int idx = lo; idx < hi; ++idx) {
                        const float tsv = P.sv_t[idx];
                        float b[7];
                        const bool at_end = (tsv == S.t);
                        rk_dense<TAB>(Q.rk, (tsv - pv.t) / dtp_, b);
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            float o = un[r];
                            if (!at_end) {
                                float acc = b[0] * k[0][r];
#pragma unroll
                                for (int j = 1; j < 7; ++j) acc += b[j] * k[j][r];
                                o = up[r] + dtp_ * acc;
                            }
                            if (valid(r)) P.sv_out[((size_t)gcol * P.nsave + idx) * P.D + feat(r)] = o;
                        }
                    }
                }
            }
        }
        if (MODE == MW_FINISH || (MODE == MW_SOLVE && S.done)) {
            if constexpr (MODE == MW_SOLVE) { if (writer) *P.ctl_final = S; }
            if (!Q.u_out) return;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (valid(r)) Q.u_out[(size_t)gcol * P.D + feat(r)] = S.live < 0 ? P.x[(size_t)gcol * P.D + feat(r)] : P.arena[(long long)S.live * P.rec_stride + L.unew() + fo + 256 * r];
            return;
        }
        if (S.done) return;
        const float t = S.t;
        const float dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
        const int rec = P.tape ? nn : (S.live == 0 ? 1 : 0);
        float* R = P.arena + (long long)rec * P.rec_stride;
        float part = 0.f, part1 = 0.f, part2 = 0.f;
        // Rolled stage loop with shifting partial sums, exactly as rnde_chain_kernel: Sa[i] = running combination of the i-th stage still to come
        float up[NR], Sa[SM - 1][NR], E[NR], un[NR], g6[NR], k6[NR];
        const float* Rl = P.arena + (long long)(S.live < 0 ? 0 : S.live) * P.rec_stride;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float k1;
            if (spec && S.live == nn - 1) { up[r] = sp_up[r]; k1 = sp_k[r]; }      // (the step starts from what attempt n - 1 wrote: already here)
            else if (S.live < 0) { up[r] = ldx(P.x, r); k1 = P.f0[fo + 256 * r]; }
            else { up[r] = Rl[L.unew() + fo + 256 * r]; k1 = Rl[L.k(NS) + fo + 256 * r]; }
            if (P.tape || P.nsave > 0) { R[L.upc() + fo + 256 * r] = up[r]; R[L.k1c() + fo + 256 * r] = k1; }
#pragma unroll
            for (int i = 0; i < SM - 1; ++i) Sa[i][r] = rk_fwd<TAB>(Q.rk, 0, i) * k1;
            E[r] = rk_bt<TAB>(Q.rk, 0) * k1;
            un[r] = up[r]; g6[r] = 0.f; k6[r] = 0.f;
        }
#pragma unroll 1
        for (int s = 1; s < NS; ++s) {   // zero-based stage: k_{s+1} = f(g_{s+1}, t + c_s dt)
            float gq[NR], kv[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) gq[r] = up[r] + dt * Sa[0][r];
            if (s == NS - 1) {
#pragma unroll
                for (int r = 0; r < NR; ++r) { un[r] = gq[r]; R[L.unew() + fo + 256 * r] = gq[r]; }
            } else if (P.tape) {
#pragma unroll
                for (int r = 0; r < NR; ++r) R[L.g(s + 1) + fo + 256 * r] = gq[r];
            }
            float* sl = (slab_tile && P.tape) ? slab_tile + (size_t)(2 + (NS - 1) * nn + (s - 1)) * Q.ev_stride : nullptr;
            MW_STAMP(2 + (s < 7 ? s : 6));
            eval(t + rk_c<TAB>(Q.rk, s) * dt, gq, kv, sl, s == 1 ? dbg : nullptr);
            if (s == 5 && P.reg_kind >= 2) {      // (stiffness estimate: 7-stage pairs only, the host refuses it otherwise)
#pragma unroll
                for (int r = 0; r < NR; ++r) { g6[r] = gq[r]; k6[r] = kv[r]; }
            }
            if (s == 6 && P.reg_kind >= 2) {   // ||k7 - k6||^2, ||unew - g6||^2
#pragma unroll
                for (int r = 0; r < NR; ++r) if (valid(r)) { const float d1 = kv[r] - k6[r], d2 = un[r] - g6[r]; part1 += d1 * d1; part2 += d2 * d2; }
            }
            const float bts = rk_bt<TAB>(Q.rk, s);
            float cs[SM - 2];
#pragma unroll
            for (int i = 0; i < SM - 2; ++i) cs[i] = rk_fwd<TAB>(Q.rk, s, i);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                R[L.k(s + 1) + fo + 256 * r] = kv[r];
                E[r] += bts * kv[r];
#pragma unroll
                for (int i = 0; i < SM - 2; ++i) Sa[i][r] = Sa[i + 1][r] + cs[i] * kv[r];
            }
        }
        // embedded error estimate, SURVEY.md B.3
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (valid(r)) {
                const float ut = dt * E[r];
                const float sk = P.abstol + fmaxf(fabsf(up[r]), fabsf(un[r])) * P.reltol;
                const float rr = ut / sk;
                part += rr * rr;
            }
        }
        MW_STAMP(9);
        part = wave_sum_f(part); part1 = wave_sum_f(part1); part2 = wave_sum_f(part2);
        if (lane == 0) { RED[wave] = part; RED[4 + wave] = part1; RED[8 + wave] = part2; }
        __syncthreads();
        MW_STAMP(10);
        if constexpr (MODE == MW_SOLVE) {
            // the meeting: wave 0 publishes this tile's three partials and collects everybody's; the sums reach the other waves through LDS
            if (wave == 0) {
                float mine[3] = {0.f, 0.f, 0.f};
                for (int w = 0; w < kMwWaves; ++w) { mine[0] += RED[w]; mine[1] += RED[4 + w]; mine[2] += RED[8 + w]; }
                double o[3];
                const bool ok = mw_exchange3(MwMeet{Q.xch, Q.abort_word, Q.epoch, Q.ntiles, Q.xch_global}, nn, mine, o, tile, lane);
                if (lane == 0) { ((double*)RED2)[0] = o[0]; ((double*)RED2)[1] = o[1]; ((double*)RED2)[2] = o[2]; RED2[6] = ok ? 1.f : 0.f; }
            }
            __syncthreads();
            if (RED2[6] == 0.f) return;                      // (a meeting timed out: abort word raised, the host redoes the solve launch by launch)
            xs[0] = ((const double*)RED2)[0]; xs[1] = ((const double*)RED2)[1]; xs[2] = ((const double*)RED2)[2];
            Sprev = S;
            __syncthreads();                                 // (RED / RED2 are rewritten by the next attempt)
        } else {
            if (tid == 0) {
                float s = 0.f, s1 = 0.f, s2 = 0.f;
                for (int w = 0; w < kMwWaves; ++w) { s += RED[w]; s1 += RED[4 + w]; s2 += RED[8 + w]; }
                float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
                ep[tile] = s; ep[P.nwg + tile] = s1; ep[2 * P.nwg + tile] = s2;
            }
            break;
        }
      }
    }
}

}  // namespace rnde
