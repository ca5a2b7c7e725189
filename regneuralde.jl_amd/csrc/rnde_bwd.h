// rnde_bwd.h -- reverse pass kernels (discretise-then-optimise).
//
// Replaces what Tracker.gradient (reference experiments/mnist_node.jl:229-232) does over the taped
// `solve` when sensealg = SensitivityADPassThrough() (reference src/models/neural_ode.jl:134): the exact
// derivative of the discrete Tsit5 program, including t, dt, EEst and the PI controller.
//
// Structure (mirror image of the forward):
//   rnde_bstep_kernel   one attempted step, reverse: the workgroup owns the same BT batch columns,
//                       the cotangents of k1..k6 and uprev live in registers for the whole attempt,
//                       J_f^T products run on the same two GEMM routines with transposed packed weights.
//   rnde_binit_kernel   reverse of k1 = f(u0,t0) and of the initial-step heuristic (2 phases).
//   rnde_wgrad_kernel   parameter gradient as two large batched GEMMs over ALL f evaluations of the
//                       solve (K = evaluations x batch), off the latency-critical sweep; deterministic
//                       (fixed-order slab reduction, no float atomics).
#pragma once
#include "rnde_fwd.h"

namespace rnde {

struct BState {  // per attempt scalar cotangents produced by that attempt's prologue
    double tb_pre, dtb_pre, qoldb, t1b, t0b, pad[3];
};
struct IBState {
    double tb, t1b, t0b, dt0b, d1b, d2b, coef_w, pad;
};
struct EvalDesc {  // one f evaluation for the parameter-gradient GEMMs
    const float* Z;  // cotangent of the layer pre-activation, M x Bpad
    const float* X;  // layer input, Nx x Bpad
    float t;
    int pad;
};

struct BwdParams {
    StepParams F;
    float* U; float* K1; float* UB1;   // running cotangents of (uprev, k1); u1bar of the initial-step heuristic
    float* zi2; float* zi1;            // [2][A], [2][HB]: z2bar/z1bar of the evaluations at (u0,t0) and (u1,t0+dt0)
    const float* svb_att;              // saveval cotangent per attempt (0 where the callback did not fire)
    BState* bstate;                    // [2]
    IBState* ibstate;                  // [2]
    float* bpart;                      // [2][nwg][4]
    float* ipart;                      // [2][nwg][4]
    float tspan_scale;                 // 0 = 1
    const float* ubar;                 // caller layout
    float* xbar;                       // caller layout
    float* tspan_out;                  // [2]
    int n_att, track_ctrl, track_initdt, reg_kind;
    const float* sv_ubar0; int sv_T;   // saveat with t0 among the save times: cotangent slice of that point (else NULL)
    int bpart_n;                       // number of per-workgroup partials the sweep wrote (differs from F.nwg when the stage engine ran it)
    const double* bsum;                // [2][4] or NULL: the three sums over bpart[parity] already formed by rnde_bpart_reduce_kernel (large batches, round 5)
};

struct BwdBuffers {
    float *U = nullptr, *K1 = nullptr, *UB1 = nullptr, *zi2 = nullptr, *zi1 = nullptr, *svb_att = nullptr;
    BState* bstate = nullptr;
    IBState* ibstate = nullptr;
    float *bpart = nullptr, *ipart = nullptr, *tspan_out = nullptr;
    EvalDesc *ev1 = nullptr, *ev2 = nullptr;
    float *slab = nullptr, *slab_r = nullptr;   // per-chunk partials; second-level partials of the two-pass reduction
    size_t slab_floats = 0;
    float *UTB = nullptr, *UNB = nullptr, *UPB0 = nullptr, *GB = nullptr;   // stage engine scratch
    EvalDesc *h_ev1 = nullptr, *h_ev2 = nullptr;  // pinned
    float* h_svb = nullptr;
    bool ready = false;
};
inline void bwd_free(BwdBuffers& b) {
    void* d[] = {b.U, b.K1, b.UB1, b.zi2, b.zi1, b.svb_att, b.bstate, b.ibstate, b.bpart, b.ipart, b.tspan_out, b.ev1, b.ev2, b.slab, b.slab_r, b.UTB, b.UNB, b.UPB0, b.GB};
    for (void* p : d) if (p) (void)hipFree(p);
    if (b.h_ev1) (void)hipHostFree(b.h_ev1);
    if (b.h_ev2) (void)hipHostFree(b.h_ev2);
    if (b.h_svb) (void)hipHostFree(b.h_svb);
    b = BwdBuffers{};
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__device__ __forceinline__ void bpart_request(const BwdParams& Q, int m, int lane, f32x4 (&e)[4]) {
    const float* part = Q.bpart + (size_t)(m & 1) * Q.bpart_n * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = *(const f32x4*)(part + 4 * (size_t)(lane + 64 * q));   // (whole block, one base address + immediate offsets: the array is padded; entries past n are not added)
}
// e0 = bpart_request(Q, m) or nullptr (request here).  Same additions in the same order either way.
__device__ __forceinline__ void finish_attempt_scalars_from(const BwdParams& Q, int m, int lane, const f32x4 (*e0)[4], double& tb, double& dtpb,
                                                            double& qoldb, double& t1b, double& t0b) {
    const BState b = Q.bstate[m & 1];
    const StepMeta mm = Q.F.meta[m];
    const float* part = Q.bpart + (size_t)(m & 1) * Q.bpart_n * 4;
    double S = 0, tau = 0, ctau = 0;
    for (int base = 0; base < Q.bpart_n; base += 256) {   // four entries per lane in flight (see sum_partials)
        f32x4 e[4];
        if (base == 0 && e0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = (*e0)[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = *(const f32x4*)(part + 4 * (size_t)(base + lane + 64 * q));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (base + lane + 64 * q < Q.bpart_n) { S += (double)e[q][0]; tau += (double)e[q][1]; ctau += (double)e[q][2]; }
    }
    S = wave_sum_d(S); tau = wave_sum_d(tau); ctau = wave_sum_d(ctau);
    const double dtb = b.dtb_pre + S / (double)mm.dt + ctau;
    double tbx = b.tb_pre + tau;
    t1b = b.t1b; t0b = b.t0b;
    if (mm.flags & F_CLAMP) { t1b += dtb; tbx -= dtb; dtpb = 0; } else dtpb = dtb;
    tb = tbx; qoldb = b.qoldb;
}
// the same tail with the three sums already formed (a kernel that reverses several attempts carries b and mm in registers: rnde_bchainmw.h SWEEP)
__device__ __forceinline__ void finish_attempt_scalars_sums(const BState& b, const StepMeta& mm, double S, double tau, double ctau, double& tb, double& dtpb,
                                                            double& qoldb, double& t1b, double& t0b) {
    const double dtb = b.dtb_pre + S / (double)mm.dt + ctau;
    double tbx = b.tb_pre + tau;
    t1b = b.t1b; t0b = b.t0b;
    if (mm.flags & F_CLAMP) { t1b += dtb; tbx -= dtb; dtpb = 0; } else dtpb = dtb;
    tb = tbx; qoldb = b.qoldb;
}
__device__ __forceinline__ void finish_attempt_scalars(const BwdParams& Q, int m, int lane, double& tb, double& dtpb,
                                                       double& qoldb, double& t1b, double& t0b) {
    finish_attempt_scalars_from(Q, m, lane, nullptr, tb, dtpb, qoldb, t1b, t0b);
}

// Large batches (round 5): every workgroup of the stage engine's reverse attempt kernel used to sum ALL bpart_n partials of attempt n + 1 in its own
// START -- 1,792 entries at B = 4096, seven dependent 256-entry blocks of cold loads in front of everything else, in each of 1,792 workgroups:
// 24 us of a 242 us reversed attempt (profiles/r05_rev_attempt_ablation.csv, `noscalar` at B = 4096).  One wave forms the three sums once, behind
// the launch that wrote the partials, in exactly the order finish_attempt_scalars_from adds them (bit-identical), and START reads three doubles.
static __global__ __launch_bounds__(64) void rnde_bpart_reduce_kernel(const BwdParams Q, int m, double* __restrict__ out) {
    const int lane = threadIdx.x;
    const float* part = Q.bpart + (size_t)(m & 1) * Q.bpart_n * 4;
    double S = 0, tau = 0, ctau = 0;
    for (int base = 0; base < Q.bpart_n; base += 256) {
        f32x4 e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = *(const f32x4*)(part + 4 * (size_t)(base + lane + 64 * q));
#pragma unroll
        for (int q = 0; q < 4; ++q) if (base + lane + 64 * q < Q.bpart_n) { S += (double)e[q][0]; tau += (double)e[q][1]; ctau += (double)e[q][2]; }
    }
    S = wave_sum_d(S); tau = wave_sum_d(tau); ctau = wave_sum_d(ctau);
    if (lane == 0) { double* o = out + 4 * (m & 1); o[0] = S; o[1] = tau; o[2] = ctau; o[3] = 0; }
}

static __global__ __launch_bounds__(64) void rnde_bfin_kernel(const BwdParams Q) {
    const int lane = threadIdx.x;
    const IBState ib = Q.ibstate[1];
    double tau0 = 0;
    for (int i = lane; i < Q.F.nwg; i += 64) tau0 += (double)Q.ipart[((size_t)Q.F.nwg + i) * 4];
    tau0 = wave_sum_d(tau0);
    if (lane == 0) {
        // (coupled controller: every rank holds the cotangent of the shared tspan summed over all shards and scaled by the world size, see
        //  rnde_node_set_coupling; tspan_scale = 1 / world makes the average over the ranks the single-device value, as for p-bar)
        const double sc = Q.tspan_scale != 0.f ? (double)Q.tspan_scale : 1.0;
        Q.tspan_out[0] = (float)(sc * (ib.t0b + tau0 + ib.tb));
        Q.tspan_out[1] = (float)(sc * ib.t1b);
    }
}

// ------------------------------------------------------------------------------------------
// Parameter gradient of one Dense layer over ALL f evaluations of the solve:
//   Wext_bar[M][Nx+2] = sum_e  Z_e[M x Bpad] * [X_e ; t_e ; 1]^T            (bias = last column)
// which is exactly the Flux.destructure segment [vec(W) (M x (Nx+1)); b (M)] in column-major order.
// One wave computes MB x NB tiles of 32x32 with v_mfma_f32_32x32x2_f32 over a chunk of evaluations and
// writes its partial to a slab; rnde_wgrad_reduce sums the slabs in fixed order.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c);   // rnde_stage.h

template <int MB, int NB>
__global__ __launch_bounds__(64) void rnde_wgrad_kernel(const EvalDesc* __restrict__ evals, int n_evals, int per_chunk,
                                                        int M, int Nx, int Bpad, float* __restrict__ slab) {
    const int lane = threadIdx.x, l31 = lane & 31, kk = lane >> 5;
    const int mtiles = (M + 31) / 32, ntiles = (Nx + 2 + 31) / 32;
    const int mblocks = (mtiles + MB - 1) / MB;
    const int mb = blockIdx.x % mblocks, nb = blockIdx.x / mblocks;
    const int chunk = blockIdx.y;
    const int e0 = chunk * per_chunk, e1 = min(n_evals, e0 + per_chunk);
    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int mrow[MB], nrow[NB];
#pragma unroll
    for (int a = 0; a < MB; ++a) mrow[a] = (mb * MB + a) * 32 + l31;
#pragma unroll
    for (int b = 0; b < NB; ++b) nrow[b] = (nb * NB + b) * 32 + l31;
    for (int e = e0; e < e1; ++e) {
        const float* __restrict__ Z = evals[e].Z;
        const float* __restrict__ X = evals[e].X;
        const float te = evals[e].t;
        float bconst[NB];  // value for the synthetic rows (time, bias), else NaN marker unused
#pragma unroll
        for (int b = 0; b < NB; ++b) bconst[b] = nrow[b] == Nx ? te : (nrow[b] == Nx + 1 ? 1.f : 0.f);
        // batches of KB column pairs: all operand loads of a batch are issued before its MFMAs so that ~50 loads are
        // in flight per wave (the loop was load-latency bound with 2 pairs in flight: 28 TF)
        constexpr int KB = 8;
        for (int c = 0; c < Bpad; c += 2 * KB) {
            float av[KB][MB], bv[KB][NB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int cc = c + 2 * u + kk;
                const bool cok = cc < Bpad;
#pragma unroll
                for (int a = 0; a < MB; ++a) av[u][a] = (cok && mrow[a] < M) ? Z[(size_t)cc * M + mrow[a]] : 0.f;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) bv[u][bb] = cok ? (nrow[bb] < Nx ? X[(size_t)cc * Nx + nrow[bb]] : bconst[bb]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][bb], acc[a][bb], 0, 0, 0);
        }
    }
    float* out = slab + (size_t)chunk * M * (Nx + 2);
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ncol = (nb * NB + b) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (mb * MB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (mr < M && ncol < Nx + 2) out[(size_t)ncol * M + mr] = acc[a][b][r];
            }
        }
}

// ---- LDS-staged variant (used by default) -----------------------------------------------------------------
// A workgroup of 4 waves computes a 128 x 128 block of Wext_bar for its chunk of evaluations: 32 batch columns of Z
// (128 rows) and of [X; t; 1] (128 rows) are staged per step with 16-byte coalesced loads, then every wave runs
// 16 K-pairs x 4 tiles of v_mfma_f32_32x32x2_f32 from LDS.  SPLIT_M: the workgroups tile the M side (layer 2,
// M = D) and each wave owns one 32-row M tile against all 4 N tiles; otherwise they tile the N side (layer 1,
// N = D + 2) and each wave owns one N tile against all 4 M tiles.  Output: same slab format as rnde_wgrad_kernel.
template <bool SPLIT_M>
__global__ __launch_bounds__(256) void rnde_wgrad2_kernel(const EvalDesc* __restrict__ evals, int n_evals, int per_chunk,
                                                          int M, int Nx, int Bpad, float* __restrict__ slab) {
    constexpr int KC = 32;
    __shared__ __attribute__((aligned(16))) float Zl[KC][128];
    __shared__ __attribute__((aligned(16))) float Xl[KC][128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, kk = lane >> 5;
    const int blk = blockIdx.x, chunk = blockIdx.y;
    const int m0 = SPLIT_M ? blk * 128 : 0, n0 = SPLIT_M ? 0 : blk * 128;
    // chunk = a contiguous range of 32-column steps of the (evaluation, column) axis -- not whole evaluations -- so that the
    // host can pick the workgroup count that fills the chip in whole rounds (per_chunk = steps per chunk)
    const int steps_per_eval = (Bpad + 32 - 1) / 32;
    const int s_lo = chunk * per_chunk, s_hi = min(n_evals * steps_per_eval, s_lo + per_chunk);
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const bool mvec = (M & 3) == 0, nvec = (Nx & 3) == 0;
    // software pipeline over (evaluation, 32-column step): the next step's 8 float4 are fetched into registers while
    // the current step's 64 MFMAs per wave run, and written to LDS after the barrier that retires the current step
    const int total_steps = max(0, s_hi - s_lo);
    f32x4 zreg[4], xreg[4];
    auto fetch = [&](int step) {
        const int e = (s_lo + step) / steps_per_eval, c0 = ((s_lo + step) % steps_per_eval) * KC;
        const float* __restrict__ Z = evals[e].Z;
        const float* __restrict__ X = evals[e].X;
        const float te = evals[e].t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, col = idx >> 5, r4 = (idx & 31) * 4;
            const int cc = c0 + col;
            f32x4 zv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
            if (cc < Bpad) {
                const int mr = m0 + r4, nr = n0 + r4;
                if (mvec && mr + 3 < M) zv = *(const f32x4*)(Z + (size_t)cc * M + mr);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (mr + i < M) zv[i] = Z[(size_t)cc * M + mr + i];
                }
                if (nvec && nr + 3 < Nx) xv = *(const f32x4*)(X + (size_t)cc * Nx + nr);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = nr + i;
                        xv[i] = n < Nx ? X[(size_t)cc * Nx + n] : (n == Nx ? te : (n == Nx + 1 ? 1.f : 0.f));
                    }
                }
            }
            zreg[q] = zv; xreg[q] = xv;
        }
    };
    if (total_steps > 0) fetch(0);
    for (int step = 0; step < total_steps; ++step) {
        __syncthreads();                       // every wave is done reading the previous step's LDS image
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, col = idx >> 5, r4 = (idx & 31) * 4;
            *(f32x4*)&Zl[col][r4] = zreg[q];
            *(f32x4*)&Xl[col][r4] = xreg[q];
        }
        __syncthreads();
        if (step + 1 < total_steps) fetch(step + 1);   // in flight under the MFMAs below
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const int col = 2 * kp + kk;
            if (SPLIT_M) {
                const float a = Zl[col][32 * w + l31];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xl[col][32 * t + l31], acc[t], 0, 0, 0);
            } else {
                const float bq = Xl[col][32 * w + l31];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Zl[col][32 * t + l31], bq, acc[t], 0, 0, 0);
            }
        }
    }
    float* out = slab + (size_t)chunk * M * (Nx + 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int mt = SPLIT_M ? w : t, nt = SPLIT_M ? t : w;
        const int ncol = n0 + 32 * nt + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = m0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (mr < M && ncol < Nx + 2) out[(size_t)ncol * M + mr] = acc[t][r];
        }
    }
}

// ---- 16x16x4 variant with near-zero tile padding (used when both widths are multiples of 4, wide side <= 800, narrow <= 112) ----
// rnde_wgrad2_kernel pads 102 -> 128 and 784 -> 896 (30 % of its MFMAs multiply zeros) and re-reads the narrow operand once per
// 128-row block (7x).  Here the wide side (784 rows of z2bar, or the 786 rows of [g; t; 1]) is split into TWO halves of 16-row
// tiles (25 + 24, or 25 + 25), a workgroup of 7 waves owns one half against the whole narrow side (7 tiles of 16 = 112 >= 102),
// wave w taking the tiles w, w + 7, w + 14, (w + 21): 28 accumulator tiles per wave at most.  Per 32-column step the half-tile
// of the wide operand (<= 400 x 32) and the narrow operand (112 x 32) are brought into LDS by `global_load_lds` one step ahead; LDS row strides are = 16 (mod 64) floats, so the four 16-lane groups of an MFMA operand read
// hit distinct bank windows.  TALL_IS_Z: the wide operand is Z (layer 2) or [X; t; 1] (layer 1).
#ifndef RNDE_WGRAD3_NT
#define RNDE_WGRAD3_NT 1   // non-temporal wide-operand stream: the reverse sweep running beside the side-stream launches keeps its L2 (32.55 -> 32.23 us per reversed attempt)
#endif
template <bool TALL_IS_Z>
__global__ __launch_bounds__(448) void rnde_wgrad3_kernel(const EvalDesc* __restrict__ evals, int n_evals, int per_chunk,
                                                          int M, int Nx, int Bpad, float* __restrict__ slab) {
    constexpr int KC = 32, TLS = 464, SLS = 144;           // LDS row strides (floats), both = 16 (mod 64)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* Tl = wsm;                      // [2][KC][TLS] interleaved with
    float* Sl = wsm + KC * TLS;           // [2][KC][SLS]: buffer b at + b * KC * (TLS + SLS)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow = lane & 15, kk = lane >> 4;
    const int TR = TALL_IS_Z ? M : Nx + 2, SR = TALL_IS_Z ? Nx + 2 : M;
    const int TT = (TR + 15) >> 4, T0 = (TT + 1) >> 1;
    const int half = blockIdx.x, chunk = blockIdx.y;
    const int tile_lo = half ? T0 : 0, tile_hi = half ? TT : T0, ntile = tile_hi - tile_lo;     // <= 25
    const int row_lo = 16 * tile_lo, nrow = 16 * ntile;                                           // <= 400
    const int steps_per_eval = (Bpad + KC - 1) / KC;
    const int s_lo = chunk * per_chunk, s_hi = min(n_evals * steps_per_eval, s_lo + per_chunk);
    const int total_steps = max(0, s_hi - s_lo);
    f32x4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 7; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool has3 = w + 21 < ntile;                      // tiles w, w + 7, w + 14 always exist (host: both halves >= 21 tiles)
    const int TRp = TALL_IS_Z ? M : Nx, SRp = TALL_IS_Z ? Nx : M;
    // Staging of one 32-column step WITHOUT registers: `global_load_lds_dwordx4` moves 64 lanes x 16 bytes from global memory
    // into 1 KiB of consecutive LDS.  The wide half of a column (<= 400 rows) is two such units (rows 0..255, 256..), two columns of
    // the narrow operand (2 x 144 floats apart) are one; 64 + 16 units per step, dealt to the 7 waves.  Lanes whose float4 has no
    // source (padding rows of the last tile, the synthetic rows) are masked off: the whole LDS image is zeroed once, and the
    // synthetic {t, 1, 0, 0} rows of [X; t; 1] -- one float4 per column -- are written by 32 threads per step.  M and Nx are
    // multiples of 4, so a float4 never mixes kinds.  (The register-staged form of this kernel spent ~330 instructions per wave
    // and step behind the last MFMA on index arithmetic, selects and LDS stores, and held 40 registers for data in flight.)
    for (int i = tid; i < 2 * KC * (TLS + SLS) / 4; i += 448) ((f32x4*)wsm)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wide_syn = (!TALL_IS_Z && TRp >= row_lo && TRp < row_lo + nrow) ? TRp - row_lo : -1;   // row (in this half) of {t, 1, 0, 0}
    const int narrow_syn = TALL_IS_Z ? SRp : -1;
    // Units are dealt statically: wide unit w + 7 j (j < 10, 64 in all: column (w + 7 j) / 2, rows 0.. or 256..), narrow unit w + 7 j
    // (j < 3, 16 in all: columns 2 u and 2 u + 1).  Per slot the lane's element offset from the step's first column is fixed
    // (-1: this lane has nothing to fetch in the slot), so a step costs one DMA instruction per slot and no index arithmetic.
    int wg_off[10], ng_off[3];
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) {
        const int u = w + 7 * jj, c = u >> 1, row = 256 * (u & 1) + 4 * lane;
        wg_off[jj] = (u < 64 && row < nrow && row_lo + row < TRp) ? c * TRp + row : -1;
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int u = w + 7 * jj, cl = lane >= 36 ? 1 : 0, row = 4 * (lane - 36 * cl);
        ng_off[jj] = (u < 16 && (lane < 28 || lane >= 36) && row < SRp) ? (2 * u + cl) * SRp + row : -1;
    }
    int e_nx = s_lo / steps_per_eval, cs_nx = s_lo - e_nx * steps_per_eval;   // (evaluation, 32-column step in it) of the next step to stage
    auto stage_dma = [&](int buf) {                        // the next step's operands -> LDS buffer `buf`, asynchronously (vmcnt)
        const int e = e_nx, c0 = cs_nx * KC;
        if (++cs_nx == steps_per_eval) { cs_nx = 0; ++e_nx; }
        const float te = evals[e].t;
        const float* Tp = (TALL_IS_Z ? evals[e].Z : evals[e].X) + (size_t)c0 * TRp + row_lo;
        const float* Sp = (TALL_IS_Z ? evals[e].X : evals[e].Z) + (size_t)c0 * SRp;
        float* Tb = Tl + buf * (KC * (TLS + SLS));
        float* Sb = Sl + buf * (KC * (TLS + SLS));
        const int ncols = min(KC, Bpad - c0);               // (uniform) columns of the step that exist; Bpad is a multiple of 16
#pragma unroll
        for (int jj = 0; jj < 10; ++jj) {
            const int u = w + 7 * jj, c = u >> 1;
#if RNDE_WGRAD3_NT
            if (wg_off[jj] >= 0 && c < ncols) dma_unit_nt((const f32x4*)(Tp + wg_off[jj]), Tb + c * TLS + 256 * (u & 1));
#else
            if (wg_off[jj] >= 0 && c < ncols) dma_unit((const f32x4*)(Tp + wg_off[jj]), Tb + c * TLS + 256 * (u & 1));
#endif
        }
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int u = w + 7 * jj;
            if (ng_off[jj] >= 0 && 2 * u + (lane >= 36 ? 1 : 0) < ncols) dma_unit((const f32x4*)(Sp + ng_off[jj]), Sb + 2 * u * SLS);
        }
        if (tid < KC) {                                     // the synthetic rows, and zeros for columns past the batch (their old image is stale)
            const f32x4 syn = tid < ncols ? (f32x4){te, 1.f, 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (wide_syn >= 0) *(f32x4*)(Tb + tid * TLS + wide_syn) = syn;
            if (narrow_syn >= 0) *(f32x4*)(Sb + tid * SLS + narrow_syn) = syn;
        }
        if (ncols < KC) {                                   // (only when Bpad is not a multiple of 32: last step of an evaluation)
            for (int i = tid; i < (KC - ncols) * (TLS / 4); i += 448) {
                const int c = ncols + i / (TLS / 4), r4 = i - (c - ncols) * (TLS / 4);
                if (4 * r4 != wide_syn) *(f32x4*)(Tb + c * TLS + 4 * r4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int i = tid; i < (KC - ncols) * (SLS / 4); i += 448) {
                const int c = ncols + i / (SLS / 4), r4 = i - (c - ncols) * (SLS / 4);
                if (4 * r4 != narrow_syn) *(f32x4*)(Sb + c * SLS + 4 * r4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    __syncthreads();                                        // the zeroed image is complete before any unit lands in it
    if (total_steps > 0) stage_dma(0);
    wait_vm<0>();
    __syncthreads();
    for (int step = 0; step < total_steps; ++step) {
        const float* Tb = Tl + (step & 1) * (KC * (TLS + SLS));
        const float* Sb = Sl + (step & 1) * (KC * (TLS + SLS));
        // Every wave has left the previous step (the barrier below), so the other buffer is free: the next step's units go out
        // under this step's MFMAs -- after the first k-step in waves 0..3, after the fifth in waves 4..6, so that of the two waves
        // that share a SIMD one keeps the matrix pipe fed while the other runs the ~300 scalar instructions of the staging.
        // the operands of k-step s + 1 are requested from LDS before the MFMAs of k-step s are issued (two register sets)
        float a[2][4], bq[2][7];
        auto operands = [&](int s, float (&av)[4], float (&bv)[7]) {
            const int col = 4 * s + kk;
#pragma unroll
            for (int i = 0; i < 3; ++i) av[i] = Tb[col * TLS + 16 * (w + 7 * i) + mrow];
            av[3] = Tb[col * TLS + 16 * (has3 ? w + 21 : w) + mrow];
#pragma unroll
            for (int n = 0; n < 7; ++n) bv[n] = Sb[col * SLS + 16 * n + mrow];
        };
        operands(0, a[0], bq[0]);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            __builtin_amdgcn_sched_barrier(0);             // (keeps the compiler from hoisting all eight k-steps' operand reads: 88 registers)
            if (s + 1 < KC / 4) operands(s + 1, a[(s + 1) & 1], bq[(s + 1) & 1]);
            if ((s == 1 || s == 5) && s == (w < 4 ? 1 : 5) && step + 1 < total_steps) stage_dma((step + 1) & 1);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int n = 0; n < 7; ++n) acc[i][n] = mfma16(a[s & 1][i], bq[s & 1][n], acc[i][n]);
            if (has3) {
#pragma unroll
                for (int n = 0; n < 7; ++n) acc[3][n] = mfma16(a[s & 1][3], bq[s & 1][n], acc[3][n]);
            }
        }
        wait_vm<0>();                                       // this wave's units of step + 1 have landed
        __syncthreads();
    }
    // D register q of lane l = C[wide row 16 T + 4 (l >> 4) + q][narrow row 16 n + (l & 15)]
    float* out = slab + (size_t)chunk * M * (Nx + 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (w + 7 * i < ntile) {
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                const int sr = 16 * n + mrow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tr = row_lo + 16 * (w + 7 * i) + 4 * kk + q;
                    if (tr < TR && sr < SR) out[TALL_IS_Z ? (size_t)sr * M + tr : (size_t)tr * M + sr] = acc[i][n][q];
                }
            }
        }
    }
}

// fixed-order sum of chunks [c0, c1) of the slab -> out (blockIdx.y selects the chunk group in pass 1)
static __global__ void rnde_wgrad_reduce(const float* __restrict__ slab, int n_chunks, int per_group, long long len, float* __restrict__ out) {
    const int c0 = blockIdx.y * per_group, c1 = min(n_chunks, c0 + per_group);
    float* o = out + (size_t)blockIdx.y * len;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {       // eight chunks requested before the first add (same additions, same order)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(slab + (size_t)(c + j) * len + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; c < c1; ++c) s += __builtin_nontemporal_load(slab + (size_t)c * len + i);
        o[i] = s;
    }
}

// the same for the two layers of the stage engine in ONE launch per pass (blockIdx.z = layer): four launches of ~9 us were two too many
struct ReduceJob { const float* slab; float* out; long long len; int n_chunks, per_group; };
struct ReducePair { ReduceJob j[2]; };
static __global__ void rnde_wgrad_reduce_pair(const ReducePair R) {
    const ReduceJob J = R.j[blockIdx.z];
    const int c0 = blockIdx.y * J.per_group, c1 = min(J.n_chunks, c0 + J.per_group);
    if (c0 >= c1) return;
    float* o = J.out + (size_t)blockIdx.y * J.len;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < J.len; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(J.slab + (size_t)(c + j) * J.len + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; c < c1; ++c) s += __builtin_nontemporal_load(J.slab + (size_t)c * J.len + i);
        o[i] = s;
    }
}

}  // namespace rnde
