/*
 * rnde_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See rnde_oracle.h.
 *
 * PARITY UNPINNED versus the Julia reference (no Julia, no golden vectors; see header).
 *
 * What each block follows:
 *   dynamics f            reference experiments/mnist_node.jl:41-54, src/models/basic.jl:16-23,
 *                         experiments/latent_ode.jl:113-124, src/models/neural_ode.jl:55 (dudt_)
 *   parameter layout      Flux.destructure, reference src/models/neural_ode.jl:12
 *   initial dt            OrdinaryDiffEq 5.50.0 src/initdt.jl (out-of-place), SURVEY.md B.1
 *   Tsit5 step + EEst     OrdinaryDiffEq 5.50.0 src/perform_step/low_order_rk_perform_step.jl,
 *                         DiffEqBase 6.53.4 src/calculate_residuals.jl, SURVEY.md A.1-A.2, B.2-B.3
 *   PI controller / loop  OrdinaryDiffEq 5.50.0 src/integrators/integrator_utils.jl, SURVEY.md B.4
 *   saving callback       DiffEqCallbacks 2.16.0 src/saving.jl via reference neural_ode.jl:126-127, SURVEY.md B.5
 *   saveat dense output   OrdinaryDiffEq src/dense/low_order_rk_interpolants.jl, SURVEY.md A.3, B.6
 *   reverse pass          what Tracker.gradient (reference experiments/mnist_node.jl:229) computes for
 *                         sensealg=SensitivityADPassThrough (neural_ode.jl:134): the exact derivative of
 *                         the discrete program, SURVEY.md B.8.
 */
#include "rnde_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef RNDE_F64
#define R(x) x
#define rsqrt_ sqrt
#define rtanh tanh
#define rfabs fabs
#define rpow pow
#define rlog10 log10
#else
#define R(x) x##f
#define rsqrt_ sqrtf
#define rtanh tanhf
#define rfabs fabsf
#define rpow powf
#define rlog10 log10f
#endif

/* ---------------- Tsit5 tableau ---------------- */
static const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double TS_A[7][7] = {
    {0},
    {0.161},
    {-0.008480655492356989, 0.335480655492357},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0.0}};
static const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                -0.1447110071732629,     0.5823571654525552,     -0.45808210592918697,
                                0.015151515151515152};
/* ---------------- Dormand-Prince 5(4) (DP5; Dormand & Prince 1980; dense output: Shampine 1986 as used by scipy's RK45) ----------------
 * A second 7-stage FSAL pair of order 5 behind the same step / controller / reverse code: the tableau is DATA (cfg->solver), which is
 * what lets tests/test_oracle.py pin the generic explicit-Runge-Kutta path against scipy.integrate's RK45 step for step. */
static const double DP_C[7] = {0.0, 0.2, 0.3, 0.8, 8.0 / 9.0, 1.0, 1.0};
static const double DP_A[7][7] = {
    {0},
    {1.0 / 5.0},
    {3.0 / 40.0, 9.0 / 40.0},
    {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0},
    {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0},
    {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0},
    {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0.0}};
static const double DP_BT[7] = {-71.0 / 57600.0, 0.0, 71.0 / 16695.0, -71.0 / 1920.0, 17253.0 / 339200.0, -22.0 / 525.0, 1.0 / 40.0};
/* dense output u(t + theta dt) = uprev + dt sum_i k_i sum_j P[i][j] theta^(j+1) */
static const double DP_P[7][4] = {
    {1.0, -8048581381.0 / 2820520608.0, 8663915743.0 / 2820520608.0, -12715105075.0 / 11282082432.0},
    {0.0, 0.0, 0.0, 0.0},
    {0.0, 131558114200.0 / 32700410799.0, -68118460800.0 / 10900136933.0, 87487479700.0 / 32700410799.0},
    {0.0, -1754552775.0 / 470086768.0, 14199869525.0 / 1410260304.0, -10690763975.0 / 1880347072.0},
    {0.0, 127303824393.0 / 49829197408.0, -318862633887.0 / 49829197408.0, 701980252875.0 / 199316789632.0},
    {0.0, -282668133.0 / 205662961.0, 2019193451.0 / 616988883.0, -1453857185.0 / 822651844.0},
    {0.0, 40617522.0 / 29380423.0, -110615467.0 / 29380423.0, 69997945.0 / 29380423.0}};
static const double (*tabA(int solver))[7] { return solver == 1 ? DP_A : TS_A; }
static const double* tabC(int solver) { return solver == 1 ? DP_C : TS_C; }
static const double* tabBT(int solver) { return solver == 1 ? DP_BT : TS_BT; }

/* controller constants, order 5 */
#define BETA1 ((real)(7.0 / 50.0))
#define BETA2 ((real)(2.0 / 25.0))
#define GAMMA ((real)0.9)
#define QMIN ((real)0.2)
#define QMAX ((real)10.0)
#define QOLDINIT ((real)1e-4)
/* alg_stability_size(Tsit5()) recalled as 3.5068 (reference mnist_node.jl:73) */
#define STAB_SIZE ((real)3.5068)

void orc_tableau_of(int solver, double* a, double* c, double* bt) {
    for (int s = 0; s < 7; ++s) {
        c[s] = tabC(solver)[s];
        bt[s] = tabBT(solver)[s];
        for (int j = 0; j < 7; ++j) a[s * 7 + j] = tabA(solver)[s][j];
    }
}
void orc_tableau(double* a, double* c, double* bt) { orc_tableau_of(0, a, c, bt); }
/* dense output weights b_i(theta), SURVEY.md A.3 */
void orc_dense_weights(double th, double* b) {
    double t2 = th * th;
    b[0] = -1.0530884977290216 * th * (th - 1.3299890189751412) * (t2 - 1.4364028541716351 * th + 0.7139816917074209);
    b[1] = 0.1017 * t2 * (t2 - 2.1966568338249754 * th + 1.2949852507374631);
    b[2] = 2.490627285651252793 * t2 * (t2 - 2.38535645472061657 * th + 1.57803468208092486);
    b[3] = -16.54810288924490272 * (th - 1.21712927295533244) * (th - 0.61620406037800089) * t2;
    b[4] = 47.37952196281928122 * (th - 1.203071208372362603) * (th - 0.658047292653547382) * t2;
    b[5] = -34.87065786149660974 * (th - 1.2) * (th - 0.666666666666666667) * t2;
    b[6] = 2.5 * (th - 1.0) * (th - 0.6) * t2;
}
/* d b_i / d theta by central difference of the polynomial in double (exact to ~1e-10; used only for the
 * theta cotangent of saveat points, a second-order effect) */
void orc_dense_weights_of(int solver, double th, double* b) {
    if (solver != 1) { orc_dense_weights(th, b); return; }
    for (int i = 0; i < 7; ++i) {
        double acc = 0, tp = th;
        for (int j = 0; j < 4; ++j) { acc += DP_P[i][j] * tp; tp *= th; }
        b[i] = acc;
    }
}
static void dense_weights_deriv(int solver, double th, double* db) {
    double e = 1e-6, bp[7], bm[7];
    orc_dense_weights_of(solver, th + e, bp);
    orc_dense_weights_of(solver, th - e, bm);
    for (int i = 0; i < 7; ++i) db[i] = (bp[i] - bm[i]) / (2 * e);
}

/* ---------------- dynamics ---------------- */
int orc_param_count(const orc_arch* a) {
    int n = 0;
    for (int l = 0; l < a->n_layers; ++l) n += (a->dims[l] + (a->time_dep ? 1 : 0)) * a->dims[l + 1] + a->dims[l + 1];
    return n;
}
int orc_act_rows_total(const orc_arch* a) { /* rows of stored activations per f eval: pre_act out + every layer out */
    int n = a->pre_act ? a->dims[0] : 0;
    for (int l = 0; l < a->n_layers; ++l) n += a->dims[l + 1];
    return n;
}

/* ---- summation-order modes (round 3; tests/test_gpu_replay.py) --------------------------------------------
 * At the reference's tolerance the fp32 error estimate is rounding noise, so HOW the two Dense layers are summed and how tanh
 * is rounded decide EEst, hence dt and NFE.  The default path accumulates a dot product as one sequential chain over k (what a
 * textbook CPU sgemm does).  Mode bits, process-wide (this library is test infrastructure; set before a forward):
 *   bit 0 (1): accumulate the two-layer TDChain of experiments/mnist_node.jl:41-54 in the ORDER of the device's stage engine
 *              (regneuralde.jl_amd/csrc/rnde_stage_persist.h phase B / phase D): layer 1 as R row-block partials of WT 16-wide
 *              k-blocks, each partial = two interleaved accumulators of K = 4 matrix instructions (acc0: k-steps 0, 2 of a
 *              block, acc1: k-steps 1, 3), partials added in order r = 0..R-1, then fma(w1t, t, z) + b1; layer 2 as the same
 *              two accumulators over the k-blocks of [h; t; 1] (time and bias are K columns H and H + 1).
 *              One K = 4 instruction = mfma_k4() below, the model tools/mfma_model.py fitted to tools/micro/mfma_numerics.hip.
 *   bit 1 (2): tanh by the device's formula (rnde_device.h tanh_fast: odd polynomial below 0.55, 1 - 2/(exp2(2 log2(e)|x|)+1)
 *              above), with correctly rounded exp2f / reciprocal where the device has v_exp_f32 / v_rcp_f32 + one Newton step.
 * Other shapes than the two-layer time-dependent chain ignore bit 0. */
static int g_sum_order = 0;
void orc_set_sum_order(int mode) { g_sum_order = mode; }
int orc_get_sum_order(void) { return g_sum_order; }
void orc_set_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#endif
}

#ifndef RNDE_F64
static inline float tanh_dev(float x) {
    const float ax = fabsf(x), x2 = x * x;
    float p = -0.00671552f;
    p = fmaf(p, x2, 0.02136713f);
    p = fmaf(p, x2, -0.05391917f);
    p = fmaf(p, x2, 0.13333165f);
    p = fmaf(p, x2, -0.33333332f);
    const float small = fmaf(x, x2 * p, x);
    const float L = 2.8853900817779268f;
    const float Llo = (float)(2.8853900817779268 - (double)2.8853900817779268f);
    const float yh = ax * L;
    const float yl = fmaf(ax, L, -yh) + ax * Llo;
    float e = (float)exp2((double)yh);
    e = fmaf(e, yl * 0.6931471805599453f, e);
    const float dd = e + 1.0f;
    float r = (float)(1.0 / (double)dd);
    r = fmaf(fmaf(-dd, r, 1.0f), r, r);
    float big = fmaf(-2.0f, r, 1.0f);
    big = ax > 9.1f ? 1.0f : big;
    return ax < 0.55f ? small : copysignf(big, x);
}
#define rtanh_m(x) ((g_sum_order & 2) ? tanh_dev(x) : rtanh(x))
#define rfma fmaf
void orc_tanh_dev(const float* x, float* y, int n) { for (int i = 0; i < n; ++i) y[i] = tanh_dev(x[i]); }
#else
#define rtanh_m(x) rtanh(x)
#define rfma fma
#endif

/* one K = 4 fp32 matrix instruction on n independent outputs: acc[r] += sum_{kk<4} w[kk][r] * x[kk]
 * (v_mfma_f32_16x16x4_f32; model fitted by tools/mfma_model.py: see MFMA_MODEL below). */
#ifndef MFMA_MODEL
#define MFMA_MODEL 0
#endif
static inline void mfma_k4(real* acc, const real* const w[4], const real x[4], int n) {
#if MFMA_MODEL == 0      /* four fused multiply-adds in order kk = 0..3 */
    for (int kk = 0; kk < 4; ++kk) {
        if (!w[kk]) continue;      /* k past the end of the layer: the device multiplies packed zeros (acc unchanged) */
        const real* wk = w[kk];
        const real xk = x[kk];
        for (int r = 0; r < n; ++r) acc[r] = rfma(wk[r], xk, acc[r]);
    }
#else                    /* exact sum of the four products and the accumulator, rounded once */
    for (int r = 0; r < n; ++r) {
        long double s = (long double)acc[r];
        for (int kk = 0; kk < 4; ++kk) if (w[kk]) s += (long double)w[kk][r] * (long double)x[kk];
        acc[r] = (real)s;
    }
#endif
}

/* stage-engine geometry of the device (rnde.hip: waves per row block chosen to minimise tile padding, more waves preferred) */
static void stage_geometry(int D, int* WT, int* R) {
    int MT = (D + 15) / 16, best = 1, bw = 1 << 30;
    int lo = MT < 4 ? MT : 4;
    if (lo < 1) lo = 1;
    for (int wt = (MT < 8 ? MT : 8); wt >= lo; --wt) {
        int r = (MT + wt - 1) / wt, waste = r * wt - MT;
        if (waste < bw) { bw = waste; best = wt; }
    }
    *WT = best;
    *R = (MT + best - 1) / best;
}

/* one column of the two-layer TDChain in the device's order.  x[D], out h[H] (after tanh) and y[D]. */
static void f_col_stage_order(const orc_arch* a, const real* p, const real* x, real t, real* h, real* y, real* acc0, real* acc1) {
    const int D = a->dims[0], H = a->dims[1];
    const real* W1 = p;
    const real* W1t = W1 + (size_t)D * H;
    const real* b1 = W1t + H;
    const real* W2 = b1 + H;
    const real* W2t = W2 + (size_t)H * D;
    const real* b2 = W2t + D;
    int WT, R;
    stage_geometry(D, &WT, &R);
    const int MT = (D + 15) / 16;
    /* layer 1 */
    for (int m = 0; m < H; ++m) h[m] = 0;
    for (int rb = 0; rb < R; ++rb) {
        for (int m = 0; m < H; ++m) acc0[m] = acc1[m] = 0;
        for (int kb = 0; kb < WT; ++kb) {
            const int tile = rb * WT + kb;
            if (tile >= MT) break;
            for (int q = 0; q < 4; ++q) {
                const real* w[4];
                real xv[4];
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 16 * tile + 4 * q + kk;
                    w[kk] = k < D ? W1 + (size_t)k * H : NULL;
                    xv[kk] = k < D ? x[k] : 0;
                }
                mfma_k4((q & 1) ? acc1 : acc0, w, xv, H);
            }
        }
        for (int m = 0; m < H; ++m) h[m] += acc0[m] + acc1[m];
    }
    for (int m = 0; m < H; ++m) {
        real pre = rfma(W1t[m], t, h[m]) + b1[m];
        h[m] = a->act[0] == 1 ? rtanh_m(pre) : pre;
    }
    /* layer 2: K columns 0..H-1 hidden units, H the time column, H + 1 the bias */
    for (int r = 0; r < D; ++r) acc0[r] = acc1[r] = 0;
    const int K2b = (H + 2 + 15) / 16;
    for (int kb = 0; kb < K2b; ++kb) {
        for (int q = 0; q < 4; ++q) {
            const real* w[4];
            real xv[4];
            int any = 0;
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 16 * kb + 4 * q + kk;
                if (k < H) { w[kk] = W2 + (size_t)k * D; xv[kk] = h[k]; any = 1; }
                else if (k == H) { w[kk] = W2t; xv[kk] = t; any = 1; }
                else if (k == H + 1) { w[kk] = b2; xv[kk] = 1; any = 1; }
                else { w[kk] = NULL; xv[kk] = 0; }
            }
            if (any) mfma_k4((q & 1) ? acc1 : acc0, w, xv, D);
        }
    }
    for (int r = 0; r < D; ++r) {
        real v = acc0[r] + acc1[r];
        y[r] = a->act[1] == 1 ? rtanh_m(v) : v;
    }
}

/* forward f. acts (optional) receives [pre_act output (if any); y_1; ...; y_L], each (rows x B) col-major
 * stacked as separate blocks: block offset = rows_before * B. out = y_L.
 * Default path: ORC_CB columns at a time so that a weight row is read once per block (the per-element association order is
 * unchanged: 0 + sum_i W[:,i] x_i in ascending i, then the time column, then the bias -- results are bit-identical to a
 * column-at-a-time loop). */
#define ORC_CB 8
#define ORC_RB 128
void orc_f_forward(const orc_arch* a, const real* p, const real* u, int B, real t, real* out, real* acts) {
    int maxd = 0;
    for (int l = 0; l <= a->n_layers; ++l)
        if (a->dims[l] > maxd) maxd = a->dims[l];
    if ((g_sum_order & 1) && a->n_layers == 2 && a->time_dep && !a->pre_act) {
        const int D = a->dims[0], H = a->dims[1];
#pragma omp parallel
        {
            real* hh = (real*)malloc(sizeof(real) * (size_t)(H + 2 * maxd));
            real* a0 = hh + H;
            real* a1 = a0 + maxd;
#pragma omp for schedule(static)
            for (int c = 0; c < B; ++c) {
                real* y = out + (size_t)c * D;
                f_col_stage_order(a, p, u + (size_t)c * D, t, hh, y, a0, a1);
                if (acts) {
                    memcpy(acts + (size_t)c * H, hh, sizeof(real) * H);
                    memcpy(acts + (size_t)H * B + (size_t)c * D, y, sizeof(real) * D);
                }
            }
            free(hh);
        }
        return;
    }
    const int FCB = B >= 128 ? 16 : 8;      /* columns per block: wider blocks reuse a weight row more, narrower ones keep small batches parallel */
    const int nblk = (B + FCB - 1) / FCB;
#pragma omp parallel
    {
        real* xa = (real*)malloc(sizeof(real) * (size_t)maxd * FCB);
        real* xb = (real*)malloc(sizeof(real) * (size_t)maxd * FCB);
#pragma omp for schedule(static)
        for (int blk = 0; blk < nblk; ++blk) {
            const int c0 = blk * FCB, nc = (B - c0 < FCB) ? B - c0 : FCB;
            const real* x = u + (size_t)c0 * a->dims[0];     /* nc columns, stride xs */
            size_t xs = a->dims[0];
            int off_rows = 0;
            if (a->pre_act) {
                for (int c = 0; c < nc; ++c) {
                    for (int i = 0; i < a->dims[0]; ++i) xa[(size_t)c * maxd + i] = rtanh_m(x[(size_t)c * xs + i]);
                    if (acts) memcpy(acts + (size_t)off_rows * B + (size_t)(c0 + c) * a->dims[0], xa + (size_t)c * maxd, sizeof(real) * a->dims[0]);
                }
                off_rows += a->dims[0];
                x = xa; xs = maxd;
            }
            const real* pl = p;
            real* cur = xb;
            for (int l = 0; l < a->n_layers; ++l) {
                int in = a->dims[l], o = a->dims[l + 1], ine = in + (a->time_dep ? 1 : 0);
                const real* W = pl;
                const real* b = pl + (size_t)ine * o;
                for (int r0 = 0; r0 < o; r0 += ORC_RB) {
                    const int nr = (o - r0 < ORC_RB) ? o - r0 : ORC_RB;
                    {
                        for (int c = 0; c < nc; ++c)
                            for (int r = 0; r < nr; ++r) cur[(size_t)c * maxd + r0 + r] = 0;
                        for (int i = 0; i < in; ++i) {
                            const real* Wi = W + (size_t)i * o + r0;
                            for (int c = 0; c < nc; ++c) {
                                const real xi = x[(size_t)c * xs + i];
                                real* cc = cur + (size_t)c * maxd + r0;
                                for (int r = 0; r < nr; ++r) cc[r] += Wi[r] * xi;
                            }
                        }
                    }
                    for (int c = 0; c < nc; ++c) {
                        real* cc = cur + (size_t)c * maxd + r0;
                        if (a->time_dep) {
                            const real* Wt = W + (size_t)in * o + r0;
                            for (int r = 0; r < nr; ++r) cc[r] += Wt[r] * t;
                        }
                        for (int r = 0; r < nr; ++r) cc[r] += b[r0 + r];
                        if (a->act[l] == 1)
                            for (int r = 0; r < nr; ++r) cc[r] = rtanh_m(cc[r]);
                    }
                }
                if (acts)
                    for (int c = 0; c < nc; ++c) memcpy(acts + (size_t)off_rows * B + (size_t)(c0 + c) * o, cur + (size_t)c * maxd, sizeof(real) * o);
                off_rows += o;
                pl += (size_t)ine * o + o;
                x = cur; xs = maxd;
                cur = (cur == xb) ? xa : xb;
            }
            for (int c = 0; c < nc; ++c)
                memcpy(out + (size_t)(c0 + c) * a->dims[a->n_layers], x + (size_t)c * xs, sizeof(real) * a->dims[a->n_layers]);
        }
        free(xa);
        free(xb);
    }
}
void orc_f_eval(const orc_arch* a, const real* p, const real* u, int B, real t, real* out) {
    orc_f_forward(a, p, u, B, t, out, NULL);
}

/* reverse of f at input u (stored), activations acts (stored):
 * given kbar (D x B): ubar_out (D x B, overwritten), pbar += , returns tbar contribution. */
real orc_f_backward(const orc_arch* a, const real* p, const real* u, const real* acts, int B, real t,
                       const real* kbar, real* ubar_out, real* pbar) {
    int L = a->n_layers;
    int maxd = 0;
    for (int l = 0; l <= L; ++l)
        if (a->dims[l] > maxd) maxd = a->dims[l];
    /* per-layer offsets */
    size_t poff[ORC_MAX_LAYERS];
    int aoff[ORC_MAX_LAYERS + 1]; /* row offsets into acts: aoff[l] = offset of y_l (l=1..L); input x_0 separately */
    {
        size_t po = 0;
        int ro = a->pre_act ? a->dims[0] : 0;
        for (int l = 0; l < L; ++l) {
            poff[l] = po;
            po += (size_t)(a->dims[l] + (a->time_dep ? 1 : 0)) * a->dims[l + 1] + a->dims[l + 1];
            aoff[l + 1] = ro;
            ro += a->dims[l + 1];
        }
    }
    /* zbar per layer for all columns, kept (o x B) for the weight gradient.  ORC_CB columns at a time with zbar transposed to
     * [r][c], so that the dot products over r run as ORC_CB independent chains side by side (vector over c): every sum keeps
     * its ascending-r order, results are bit-identical to a column-at-a-time loop. */
    real* zb_all[ORC_MAX_LAYERS];
    for (int l = 0; l < L; ++l) zb_all[l] = (real*)malloc(sizeof(real) * (size_t)a->dims[l + 1] * B);
    double tbar_acc = 0;
    const int nblk = (B + ORC_CB - 1) / ORC_CB;
    double* tcol = (double*)calloc((size_t)B, sizeof(double));     /* per-column time cotangent, summed in column order below */
#pragma omp parallel
    {
        real* ga = (real*)malloc(sizeof(real) * (size_t)maxd * ORC_CB);
        real* gb = (real*)malloc(sizeof(real) * (size_t)maxd * ORC_CB);
        real* zt = (real*)malloc(sizeof(real) * (size_t)maxd * ORC_CB);
#pragma omp for schedule(static)
        for (int blk = 0; blk < nblk; ++blk) {
            const int c0 = blk * ORC_CB, nc = (B - c0 < ORC_CB) ? B - c0 : ORC_CB;
            const real* g = kbar + (size_t)c0 * a->dims[L];
            size_t gs = a->dims[L];
            real* nxt = ga;
            for (int l = L - 1; l >= 0; --l) {
                int in = a->dims[l], o = a->dims[l + 1];
                const real* W = p + poff[l];
                for (int c = 0; c < ORC_CB; ++c) {
                    if (c < nc) {
                        const real* y = acts + (size_t)aoff[l + 1] * B + (size_t)(c0 + c) * o;
                        real* zb = zb_all[l] + (size_t)(c0 + c) * o;
                        const real* gc = g + (size_t)c * gs;
                        for (int r = 0; r < o; ++r) { zb[r] = a->act[l] == 1 ? gc[r] * (1 - y[r] * y[r]) : gc[r]; zt[(size_t)r * ORC_CB + c] = zb[r]; }
                    } else {
                        for (int r = 0; r < o; ++r) zt[(size_t)r * ORC_CB + c] = 0;
                    }
                }
                int i = 0;
                for (; i + 4 <= in; i += 4) {      /* four input rows at a time: four independent chains per column keep the FMA pipes busy */
                    const real* Wi = W + (size_t)i * o;
                    real s0[ORC_CB], s1[ORC_CB], s2[ORC_CB], s3[ORC_CB];
                    for (int c = 0; c < ORC_CB; ++c) s0[c] = s1[c] = s2[c] = s3[c] = 0;
                    for (int r = 0; r < o; ++r) {
                        const real w0 = Wi[r], w1 = Wi[o + r], w2 = Wi[2 * (size_t)o + r], w3 = Wi[3 * (size_t)o + r];
                        const real* z = zt + (size_t)r * ORC_CB;
                        for (int c = 0; c < ORC_CB; ++c) { s0[c] += w0 * z[c]; s1[c] += w1 * z[c]; s2[c] += w2 * z[c]; s3[c] += w3 * z[c]; }
                    }
                    for (int c = 0; c < nc; ++c) {
                        real* d = nxt + (size_t)c * maxd + i;
                        d[0] = s0[c]; d[1] = s1[c]; d[2] = s2[c]; d[3] = s3[c];
                    }
                }
                for (; i < in; ++i) {
                    const real* Wi = W + (size_t)i * o;
                    real sacc[ORC_CB];
                    for (int c = 0; c < ORC_CB; ++c) sacc[c] = 0;
                    for (int r = 0; r < o; ++r) {
                        const real w = Wi[r];
                        const real* z = zt + (size_t)r * ORC_CB;
                        for (int c = 0; c < ORC_CB; ++c) sacc[c] += w * z[c];
                    }
                    for (int c = 0; c < nc; ++c) nxt[(size_t)c * maxd + i] = sacc[c];
                }
                if (a->time_dep) {
                    const real* Wt = W + (size_t)in * o;
                    real sacc[ORC_CB];
                    for (int c = 0; c < ORC_CB; ++c) sacc[c] = 0;
                    for (int r = 0; r < o; ++r) {
                        const real w = Wt[r];
                        const real* z = zt + (size_t)r * ORC_CB;
                        for (int c = 0; c < ORC_CB; ++c) sacc[c] += w * z[c];
                    }
                    for (int c = 0; c < nc; ++c) tcol[c0 + c] += (double)sacc[c];
                }
                g = nxt; gs = maxd;
                nxt = (nxt == ga) ? gb : ga;
            }
            for (int c = 0; c < nc; ++c) {
                real* uo = ubar_out + (size_t)(c0 + c) * a->dims[0];
                const real* gc = g + (size_t)c * gs;
                if (a->pre_act) {
                    const real* x0 = acts + (size_t)(c0 + c) * a->dims[0];
                    for (int i = 0; i < a->dims[0]; ++i) uo[i] = gc[i] * (1 - x0[i] * x0[i]);
                } else {
                    for (int i = 0; i < a->dims[0]; ++i) uo[i] = gc[i];
                }
            }
        }
        free(ga);
        free(gb);
        free(zt);
    }
    for (int c = 0; c < B; ++c) tbar_acc += tcol[c];      /* (fixed order: independent of the thread count) */
    free(tcol);
    /* weight gradients: Wbar[:,i] += sum_c zbar[:,c] * x_l[i,c] ; bbar += rowsum(zbar).  Eight input rows at a time share a pass
     * over zbar (every element still sums over c in ascending order). */
    for (int l = 0; l < L; ++l) {
        int in = a->dims[l], o = a->dims[l + 1], ine = in + (a->time_dep ? 1 : 0);
        real* Wb = pbar + poff[l];
        real* bb = Wb + (size_t)ine * o;
        const real* zb = zb_all[l];
        const real* xin; /* input of layer l: (in x B) */
        if (l == 0)
            xin = a->pre_act ? acts : u;
        else
            xin = acts + (size_t)aoff[l] * B;
        const int IB = 8, nib = (in + IB - 1) / IB;
#pragma omp parallel for schedule(static)
        for (int ib = 0; ib < nib + 1; ++ib) {
            if (ib < nib) {
                const int i0 = ib * IB, ni = (in - i0 < IB) ? in - i0 : IB;
                for (int c = 0; c < B; ++c) {
                    const real* z = zb + (size_t)c * o;
                    const real* xc = xin + (size_t)c * in + i0;
                    for (int ii = 0; ii < ni; ++ii) {
                        const real xi = xc[ii];
                        real* Wbi = Wb + (size_t)(i0 + ii) * o;
                        for (int r = 0; r < o; ++r) Wbi[r] += z[r] * xi;
                    }
                }
            } else {
                if (a->time_dep) {
                    real* Wbi = Wb + (size_t)in * o;
                    for (int c = 0; c < B; ++c) {
                        const real* z = zb + (size_t)c * o;
                        for (int r = 0; r < o; ++r) Wbi[r] += z[r] * t;
                    }
                }
                for (int c = 0; c < B; ++c) {
                    const real* z = zb + (size_t)c * o;
                    for (int r = 0; r < o; ++r) bb[r] += z[r];
                }
            }
        }
    }
    for (int l = 0; l < L; ++l) free(zb_all[l]);
    return (real)tbar_acc;
}

/* ---------------- helpers ---------------- */
static real rms_ratio(const real* a, const real* sk, size_t n) { /* sqrt(mean((a/sk)^2)) */
    double s = 0;
#ifdef RNDE_F64
    for (size_t i = 0; i < n; ++i) { double v = a[i] / sk[i]; s += v * v; }
    return sqrt(s / (double)n);
#else
    /* fp32 values, but the sum itself is carried in double so the oracle's norm does not depend on
     * summation order (the HIP kernel uses a fixed-order tree of fp32 partials + a double final sum). */
    for (size_t i = 0; i < n; ++i) { float v = a[i] / sk[i]; s += (double)(v * v); }
    return (float)sqrt(s / (double)n);
#endif
}

/* ---------------- tape ---------------- */
typedef struct {
    real t, dt, dtp_in;      /* dt actually used, dt proposed on entry */
    int clamped;             /* dt = t1 - t */
    real eest, q11, q, qold_in;
    int accepted, q_clamped, eest_zero, dtmax_clamped;
    real rej_m;              /* reject: m = min(1/qmin, q11/gamma) */
    int rej_m_is_q11;
    real eigen_est;
    int sv_index;            /* index in saveval of this step's callback value, -1 if none */
    real* uprev;             /* pointer (borrowed): u0 or previous accepted unew */
    real* k[7];              /* k[0] borrowed (k1), k[1..6] owned */
    real* unew;              /* owned */
    real* acts[7];           /* activations of stages 2..7 (index 1..6), owned */
    /* saveat points filled from this step */
    int nsv_pts;
    int sv_first;            /* first saveat index filled */
} attempt_rec;

typedef struct {
    orc_config cfg;
    int D, P, B, arows;
    /* tape */
    int n_att;
    attempt_rec* att;
    real *u0, *f0, *acts0; /* f(u0,t0) and its activations */
    real *u1, *f1, *acts1; /* initdt second eval */
    real t0, t1, dt0_small_or_clamped_flag;
    /* initdt record */
    real id_d0, id_d1, id_d2, id_dt0, id_dt1, id_dt;
    int id_dt0_const, id_dt0_clamped, id_sel; /* sel: 0 -> 100 dt0, 1 -> dt1, 2 -> dtmax */
    int id_dt1_const, id_max_is_d2;
    real* p;
    int nsave;
    real* saveat;
    int save_t0; /* saveat[0]==t0 saved from u0 */
    int n_saveval;
    int have_tape;
    /* replay (orc_set_replay): the next orc_forward follows a given sequence of proposed step sizes and accept
     * decisions instead of its own controller's -- parity tests run fp32 / fp64 / device along ONE sequence */
    int n_replay;
    real* replay_dtp;
    int* replay_acc;
} orc_handle;

void* orc_create(const orc_config* cfg) {
    orc_handle* h = (orc_handle*)calloc(1, sizeof(orc_handle));
    h->cfg = *cfg;
    h->D = cfg->arch.dims[0];
    h->P = orc_param_count(&cfg->arch);
    h->arows = orc_act_rows_total(&cfg->arch);
    h->att = (attempt_rec*)calloc((size_t)cfg->max_attempts + 1, sizeof(attempt_rec));
    return h;
}
static void free_tape(orc_handle* h) {
    for (int n = 0; n < h->n_att; ++n) {
        attempt_rec* a = &h->att[n];
        for (int s = 1; s < 7; ++s) { free(a->k[s]); free(a->acts[s]); a->k[s] = a->acts[s] = NULL; }
        free(a->unew);
        a->unew = NULL;
    }
    h->n_att = 0;
    free(h->u0); free(h->f0); free(h->acts0); free(h->u1); free(h->f1); free(h->acts1); free(h->p); free(h->saveat);
    h->u0 = h->f0 = h->acts0 = h->u1 = h->f1 = h->acts1 = h->p = h->saveat = NULL;
    h->have_tape = 0;
}
void orc_set_replay(void* hh, const real* dtp, const int* acc, int n) {
    orc_handle* h = (orc_handle*)hh;
    free(h->replay_dtp); free(h->replay_acc);
    h->replay_dtp = NULL; h->replay_acc = NULL; h->n_replay = 0;
    if (n <= 0) return;
    h->replay_dtp = (real*)malloc(sizeof(real) * n);
    h->replay_acc = (int*)malloc(sizeof(int) * n);
    memcpy(h->replay_dtp, dtp, sizeof(real) * n);
    memcpy(h->replay_acc, acc, sizeof(int) * n);
    h->n_replay = n;
}
void orc_destroy(void* hh) {
    orc_handle* h = (orc_handle*)hh;
    free_tape(h);
    free(h->replay_dtp); free(h->replay_acc);
    free(h->att);
    free(h);
}
static real* ralloc(size_t n) { return (real*)malloc(sizeof(real) * n); }

/* ---------------- one attempt (shared by forward and the kernel-parity entry) ---------------- */
static void attempt_stages(const orc_config* cfg, const real* p, const real* uprev, real* const k[7], real* unew,
                           real* const acts[7], int B, real t, real dt, real* eest_out, real* eigen_out) {
    const orc_arch* a = &cfg->arch;
    int D = a->dims[0];
    size_t N = (size_t)D * B;
    real* g = ralloc(N);
    real* g6 = ralloc(N);
    for (int s = 1; s < 7; ++s) { /* stage index s (0-based): computes k[s] = f(g_{s+1}) */
        real as[6];
        for (int j = 0; j < s; ++j) as[j] = (real)tabA(cfg->solver)[s][j];
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; ++i) {
            real acc = 0;
            for (int j = 0; j < s; ++j) acc += as[j] * k[j][i];
            g[i] = uprev[i] + dt * acc;
        }
        if (s == 5) memcpy(g6, g, sizeof(real) * N);
        if (s == 6) memcpy(unew, g, sizeof(real) * N);
        orc_f_forward(a, p, g, B, t + (real)tabC(cfg->solver)[s] * dt, k[s], acts ? acts[s] : NULL);
    }
    /* error estimate */
    real bt[7];
    for (int j = 0; j < 7; ++j) bt[j] = (real)tabBT(cfg->solver)[j];
    double ssum = 0;
    double* colsum = (double*)malloc(sizeof(double) * (size_t)B);     /* per-column sums, added in column order: the result does not depend on the thread count */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < B; ++c) {
        double cs = 0;
        for (size_t i = (size_t)c * D; i < (size_t)(c + 1) * D; ++i) {
            real acc = 0;
            for (int j = 0; j < 7; ++j) acc += bt[j] * k[j][i];
            real ut = dt * acc;
            real au = rfabs(uprev[i]), an = rfabs(unew[i]);
            real sk = cfg->abstol + (au > an ? au : an) * cfg->reltol;
            real r = ut / sk;
            cs += (double)(r * r);
        }
        colsum[c] = cs;
    }
    for (int c = 0; c < B; ++c) ssum += colsum[c];
    free(colsum);
    *eest_out = (real)sqrt(ssum / (double)N);
    if (eigen_out) { /* eigen_est = ||k7-k6|| / ||u - g6||   */
        double n1 = 0, n2 = 0;
        for (size_t i = 0; i < N; ++i) {
            double d1 = (double)k[6][i] - (double)k[5][i], d2 = (double)unew[i] - (double)g6[i];
            n1 += d1 * d1;
            n2 += d2 * d2;
        }
        *eigen_out = (real)(sqrt(n1) / sqrt(n2));
    }
    free(g);
    free(g6);
}

void orc_tsit5_attempt(const orc_config* cfg, const real* p, const real* uprev, const real* k1, int B, real t,
                       real dt, real* kout, real* unew, real* eest, real* eigen_est) {
    size_t N = (size_t)cfg->arch.dims[0] * B;
    real* k[7];
    k[0] = (real*)k1;
    for (int s = 1; s < 7; ++s) k[s] = kout + (size_t)(s - 1) * N;
    attempt_stages(cfg, p, uprev, k, unew, NULL, B, t, dt, eest, eigen_est);
}

/* ---------------- initial dt ---------------- */
static real initdt_impl(orc_handle* h, const orc_config* cfg, const real* p, const real* u0, int B, real t0, real t1,
                        real* f0, real* acts0, real* u1, real* f1, real* acts1) {
    const orc_arch* a = &cfg->arch;
    size_t N = (size_t)a->dims[0] * B;
    real dtmax = t1 - t0;
    real* sk = ralloc(N);
    for (size_t i = 0; i < N; ++i) sk[i] = cfg->abstol + rfabs(u0[i]) * cfg->reltol;
    real d0 = rms_ratio(u0, sk, N);
    orc_f_forward(a, p, u0, B, t0, f0, acts0);
    real d1 = rms_ratio(f0, sk, N);
    real dt0;
    int c0 = 0, cl = 0;
    if (d0 < R(1e-5) || d1 < R(1e-5)) { dt0 = R(1e-6); c0 = 1; }
    else dt0 = (d0 / d1) / R(100.0);
    if (dtmax < dt0) { dt0 = dtmax; cl = 1; }
    for (size_t i = 0; i < N; ++i) u1[i] = u0[i] + dt0 * f0[i];
    orc_f_forward(a, p, u1, B, t0 + dt0, f1, acts1);
    real* df = ralloc(N);
    for (size_t i = 0; i < N; ++i) df[i] = f1[i] - f0[i];
    real d2 = rms_ratio(df, sk, N) / dt0;
    free(df);
    free(sk);
    real m = d1 > d2 ? d1 : d2;
    real dt1;
    int c1 = 0;
    if (m <= R(1e-15)) { real a1 = R(1e-6), a2 = dt0 * R(1e-3); dt1 = a1 > a2 ? a1 : a2; c1 = 1; }
    else dt1 = (real)pow(10.0, (double)(-(R(2.0) + rlog10(m)) / R(5.0))); /* Julia: 10.0^(...) is Float64, then convert */
    real dt = R(100.0) * dt0;
    int sel = 0;
    if (dt1 < dt) { dt = dt1; sel = 1; }
    if (dtmax < dt) { dt = dtmax; sel = 2; }
    if (h) {
        h->id_d0 = d0; h->id_d1 = d1; h->id_d2 = d2; h->id_dt0 = dt0; h->id_dt1 = dt1; h->id_dt = dt;
        h->id_dt0_const = c0; h->id_dt0_clamped = cl; h->id_sel = sel; h->id_dt1_const = c1; h->id_max_is_d2 = (d2 >= d1);
    }
    return dt;
}
real orc_initdt(const orc_config* cfg, const real* p, const real* u0, int B, real t0, real t1, real* f0_out) {
    size_t N = (size_t)cfg->arch.dims[0] * B;
    real *f0 = ralloc(N), *u1 = ralloc(N), *f1 = ralloc(N);
    real dt = initdt_impl(NULL, cfg, p, u0, B, t0, t1, f0, NULL, u1, f1, NULL);
    if (f0_out) memcpy(f0_out, f0, sizeof(real) * N);
    free(f0); free(u1); free(f1);
    return dt;
}

/* callback value (reference mnist_node.jl:67, :74-79, :88-97) */
static real cb_value(const orc_config* cfg, real eest, real dt, real eigen) {
    switch (cfg->reg_kind) {
        case 1: return eest * dt;
        case 2: { real s = rfabs(eigen); return (s == 0 || isnan(s)) ? 0 : s / STAB_SIZE; }
        case 3: {
            real e = eest * dt;
            real v = (e == 0 || isnan(e)) ? 0 : e;
            real s = eigen;
            v += R(0.1) * ((s == 0 || isnan(s)) ? 0 : s / STAB_SIZE);
            return v;
        }
        default: return 0;
    }
}

/* ---------------- forward solve ---------------- */
int orc_forward(void* hh, const real* x, const real* p, int B, real t0, real t1, const real* saveat, int nsave,
                real* u_out, long* nfe, real* saveval, int* nsaveval, real* steps_log, int* nattempts) {
    orc_handle* h = (orc_handle*)hh;
    const orc_config* cfg = &h->cfg;
    const orc_arch* a = &cfg->arch;
    free_tape(h);
    int D = h->D;
    size_t N = (size_t)D * B;
    h->B = B; h->t0 = t0; h->t1 = t1;
    h->p = ralloc(h->P); memcpy(h->p, p, sizeof(real) * h->P);
    h->u0 = ralloc(N); memcpy(h->u0, x, sizeof(real) * N);
    h->f0 = ralloc(N); h->u1 = ralloc(N); h->f1 = ralloc(N);
    h->acts0 = ralloc((size_t)h->arows * B); h->acts1 = ralloc((size_t)h->arows * B);
    h->nsave = nsave;
    if (nsave) { h->saveat = ralloc(nsave); memcpy(h->saveat, saveat, sizeof(real) * nsave); }
    long nf = 0;
    real dtp = initdt_impl(h, cfg, p, h->u0, B, t0, t1, h->f0, h->acts0, h->u1, h->f1, h->acts1);
    nf += 2;
    /* fsalfirst = f(u0,t0): numerically identical to f0, counted as one more evaluation */
    nf += 1;
    const int replay = h->n_replay;
    if (replay) dtp = h->replay_dtp[0];
    real t = t0, qold = QOLDINIT, dtmax = t1 - t0;
    int nsv = 0, ret = 0, n = 0, next_save = 0;
    real dtmin = (real)(sizeof(real) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
    if (cfg->reg_kind && cfg->cb_save_start) {
        /* callback initialisation fires before the initial dt is chosen: EEst = 1, dt = 0 -> func = 0 for the
         * error estimate; eigen_est initial value 1 for the stiffness variant. */
        saveval[nsv++] = cb_value(cfg, 1, 0, 1);
    }
    h->save_t0 = 0;
    if (nsave && saveat[0] == t0) { /* save_start */
        for (int c = 0; c < B; ++c) memcpy(u_out + ((size_t)c * nsave + 0) * D, x + (size_t)c * D, sizeof(real) * D);
        next_save = 1;
        h->save_t0 = 1;
    }
    real* uprev = h->u0;
    real* k1 = h->f0;
    while (replay ? n < replay : t < t1) {
        if (n >= cfg->max_attempts) { ret = 1; break; }
        attempt_rec* r = &h->att[n];
        memset(r, 0, sizeof(*r));
        r->sv_index = -1;
        r->t = t; r->dtp_in = dtp; r->qold_in = qold;
        real dt = dtp;
        if (t1 - t < dt) { dt = t1 - t; r->clamped = 1; }
        r->dt = dt;
        if (!(dt > dtmin) || isnan(dt)) { ret = isnan(dt) ? 3 : 2; break; }
        r->uprev = uprev;
        r->k[0] = k1;
        for (int s = 1; s < 7; ++s) { r->k[s] = ralloc(N); r->acts[s] = ralloc((size_t)h->arows * B); }
        r->unew = ralloc(N);
        real eest, eig = 0;
        attempt_stages(cfg, p, uprev, r->k, r->unew, r->acts, B, t, dt, &eest, cfg->reg_kind >= 2 ? &eig : NULL);
        nf += 6;
        h->n_att = ++n;
        r->eest = eest; r->eigen_est = eig;
        if (!(eest == eest) || isinf(eest)) { ret = 3; break; }
        /* PI controller */
        real q, q11 = 0;
        if (eest == 0) { q = 1 / QMAX; r->eest_zero = 1; r->q_clamped = 1; }
        else {
            q11 = rpow(eest, BETA1);
            q = q11 / rpow(qold, BETA2);
            real qg = q / GAMMA;
            real lo = 1 / QMAX, hi = 1 / QMIN;
            if (qg < lo) { q = lo; r->q_clamped = 1; }
            else if (qg > hi) { q = hi; r->q_clamped = 1; }
            else q = qg;
        }
        r->q11 = q11; r->q = q;
        r->accepted = replay ? h->replay_acc[n - 1] : (eest <= 1);
        if (steps_log) { steps_log[4 * (n - 1) + 0] = t; steps_log[4 * (n - 1) + 1] = dt; steps_log[4 * (n - 1) + 2] = eest; steps_log[4 * (n - 1) + 3] = (real)r->accepted; }
        if (r->accepted) {
            qold = eest > QOLDINIT ? eest : QOLDINIT;
            real dtnew = dt / q;
            if (dtmax < dtnew) { dtnew = dtmax; r->dtmax_clamped = 1; }
            real tnew = t + dt;
            /* saveat points in (t, tnew] from the dense output */
            r->sv_first = next_save; r->nsv_pts = 0;
            while (next_save < nsave && saveat[next_save] <= tnew) {
                real ts = saveat[next_save];
                if (ts == tnew) {
                    for (int c = 0; c < B; ++c) memcpy(u_out + ((size_t)c * nsave + next_save) * D, r->unew + (size_t)c * D, sizeof(real) * D);
                } else {
                    double bth[7];
                    orc_dense_weights_of(cfg->solver, (double)((ts - t) / dt), bth);
                    for (int c = 0; c < B; ++c)
                        for (int i = 0; i < D; ++i) {
                            size_t e = (size_t)c * D + i;
                            real acc = 0;
                            for (int j = 0; j < 7; ++j) acc += (real)bth[j] * r->k[j][e];
                            u_out[((size_t)c * nsave + next_save) * D + i] = uprev[e] + dt * acc;
                        }
                }
                ++next_save; ++r->nsv_pts;
            }
            if (cfg->reg_kind) { r->sv_index = nsv; saveval[nsv++] = cb_value(cfg, eest, dt, eig); }
            t = tnew; dtp = dtnew; uprev = r->unew; k1 = r->k[6];
            if (replay && n < replay) dtp = h->replay_dtp[n];
        } else {
            real m = 1 / QMIN, m2 = q11 / GAMMA;
            r->rej_m_is_q11 = 0;
            if (m2 < m) { m = m2; r->rej_m_is_q11 = 1; }
            r->rej_m = m;
            dtp = dt / m;
            if (dtmax < dtp) dtp = dtmax;
            if (replay && n < replay) dtp = h->replay_dtp[n];
        }
    }
    if (!nsave) memcpy(u_out, uprev, sizeof(real) * N);
    *nfe = nf;
    *nsaveval = nsv;
    h->n_saveval = nsv;
    *nattempts = n;
    h->have_tape = (ret == 0);
    (void)a;
    return ret;
}

/* per attempt of the last forward: t, dt, dtp_in (proposed size on entry, before the clamp to t1 - t), EEst, accepted, q */
int orc_steps_ext(void* hh, real* out6, int cap) {
    orc_handle* h = (orc_handle*)hh;
    int n = h->n_att < cap ? h->n_att : cap;
    for (int i = 0; i < n; ++i) {
        const attempt_rec* r = &h->att[i];
        out6[6 * i + 0] = r->t; out6[6 * i + 1] = r->dt; out6[6 * i + 2] = r->dtp_in;
        out6[6 * i + 3] = r->eest; out6[6 * i + 4] = (real)r->accepted; out6[6 * i + 5] = r->q;
    }
    return h->n_att;
}

/* ---------------- reverse pass ---------------- */
/* Sums over the state arrays are taken in fixed chunks (per-chunk partial sums in parallel, chunks added in order), so the
 * reverse pass does not depend on the number of threads or their scheduling. */
#define ORC_CHUNK 4096
static double dotp(const real* a, const real* b, size_t n) {
    const size_t nch = (n + ORC_CHUNK - 1) / ORC_CHUNK;
    double* part = (double*)malloc(sizeof(double) * (nch ? nch : 1));
#pragma omp parallel for schedule(static)
    for (size_t ch = 0; ch < nch; ++ch) {
        const size_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
        double s = 0;
        for (size_t i = ch * ORC_CHUNK; i < i1; ++i) s += (double)a[i] * (double)b[i];
        part[ch] = s;
    }
    double s = 0;
    for (size_t ch = 0; ch < nch; ++ch) s += part[ch];
    free(part);
    return s;
}

int orc_backward(void* hh, const real* ubar, const real* svbar, real* xbar, real* pbar, real* tspanbar) {
    orc_handle* h = (orc_handle*)hh;
    if (!h->have_tape) return -1;
    const orc_config* cfg = &h->cfg;
    const orc_arch* a = &cfg->arch;
    int D = h->D, B = h->B, nsave = h->nsave;
    size_t N = (size_t)D * B;
    const real* p = h->p;
    memset(pbar, 0, sizeof(real) * h->P);
    real* U = (real*)calloc(N, sizeof(real));   /* cotangent of current uprev' */
    real* K1 = (real*)calloc(N, sizeof(real));  /* cotangent of current k1' */
    real* kb[7];
    for (int j = 0; j < 7; ++j) kb[j] = ralloc(N);
    real* unb = ralloc(N);
    real* upb = ralloc(N);
    real* gb = ralloc(N);
    real* utb = ralloc(N);
    real* gtmp = ralloc(N);
    double tb = 0, dtpb = 0, qoldb = 0, t1b = 0, t0b = 0; /* scalar cotangents carried backwards */
    if (!nsave) memcpy(U, ubar, sizeof(real) * N);
    for (int n = h->n_att - 1; n >= 0; --n) {
        attempt_rec* r = &h->att[n];
        real dt = r->dt, t = r->t;
        double eb = 0, dtb = 0, q11b = 0, qb = 0, qoldb_in = 0, tb_in = tb;
        /* ----- outputs of the attempt ----- */
        if (r->accepted) {
            /* saveat points */
            memcpy(unb, U, sizeof(real) * N);
            memset(upb, 0, sizeof(real) * N);
            for (int j = 0; j < 7; ++j) memset(kb[j], 0, sizeof(real) * N);
            memcpy(kb[6], K1, sizeof(real) * N); /* FSAL: k1' = k7 */
            for (int sidx = r->sv_first; sidx < r->sv_first + r->nsv_pts; ++sidx) {
                real ts = h->saveat[sidx];
                real tnew = t + dt;
                if (ts == tnew) {
                    for (int c = 0; c < B; ++c)
                        for (int i = 0; i < D; ++i) unb[(size_t)c * D + i] += ubar[((size_t)c * nsave + sidx) * D + i];
                } else {
                    double th = (double)((ts - t) / dt), bth[7], dbth[7];
                    orc_dense_weights_of(cfg->solver, th, bth);
                    dense_weights_deriv(cfg->solver, th, dbth);
                    double d_dt = 0, d_th = 0;
                    for (int c = 0; c < B; ++c)
                        for (int i = 0; i < D; ++i) {
                            size_t e = (size_t)c * D + i;
                            real ub = ubar[((size_t)c * nsave + sidx) * D + i];
                            upb[e] += ub;
                            double acc = 0, dacc = 0;
                            for (int j = 0; j < 7; ++j) {
                                kb[j][e] += dt * (real)bth[j] * ub;
                                acc += bth[j] * (double)r->k[j][e];
                                dacc += dbth[j] * (double)r->k[j][e];
                            }
                            d_dt += (double)ub * acc;
                            d_th += (double)ub * (double)dt * dacc;
                        }
                    dtb += d_dt;
                    /* theta = (ts - t)/dt */
                    tb_in += -d_th / (double)dt;
                    dtb += -d_th * th / (double)dt;
                }
            }
            double eigb = 0; /* cotangent of eigen_est (reference mnist_node.jl:74-79, :88-97) */
            if (r->sv_index >= 0 && svbar) {
                double sb = (double)svbar[r->sv_index];
                real eg = r->eigen_est;
                int eg_ok = !(eg == 0 || isnan(eg));
                switch (cfg->reg_kind) {
                    case 1: eb += sb * (double)dt; dtb += sb * (double)r->eest; break;
                    case 2: if (eg_ok) eigb += sb * (eg > 0 ? 1.0 : -1.0) / (double)STAB_SIZE; break;   /* stab * |eigen_est| */
                    case 3: { real e = r->eest * dt; if (!(e == 0 || isnan(e))) { eb += sb * (double)dt; dtb += sb * (double)r->eest; }
                              if (eg_ok) eigb += 0.1 * sb / (double)STAB_SIZE; } break;
                    default: break;
                }
            }
            if (eigb != 0) {
                /* eigen_est = N1/N2, N1 = ||k7-k6||, N2 = ||unew-g6||, g6 = uprev + dt sum_j a6j kj   */
                double n1 = 0, n2 = 0;
                for (size_t i = 0; i < N; ++i) {
                    real acc = 0;
                    for (int j = 0; j < 5; ++j) acc += (real)tabA(cfg->solver)[5][j] * r->k[j][i];
                    gtmp[i] = r->uprev[i] + dt * acc;
                    double d1 = (double)r->k[6][i] - (double)r->k[5][i], d2 = (double)r->unew[i] - (double)gtmp[i];
                    n1 += d1 * d1; n2 += d2 * d2;
                }
                n1 = sqrt(n1); n2 = sqrt(n2);
                if (n1 > 0 && n2 > 0) {
                    const double c1 = eigb / (n2 * n1), c2 = -eigb * (n1 / n2) / (n2 * n2);
                    double d_dt = 0;
                    for (size_t i = 0; i < N; ++i) {
                        const real d1 = r->k[6][i] - r->k[5][i], d2 = r->unew[i] - gtmp[i];
                        kb[6][i] += (real)(c1 * (double)d1);
                        kb[5][i] -= (real)(c1 * (double)d1);
                        unb[i] += (real)(c2 * (double)d2);
                        const real g6b = -(real)(c2 * (double)d2);
                        upb[i] += g6b;
                        real acc = 0;
                        for (int j = 0; j < 5; ++j) { kb[j][i] += dt * (real)tabA(cfg->solver)[5][j] * g6b; acc += (real)tabA(cfg->solver)[5][j] * r->k[j][i]; }
                        d_dt += (double)g6b * (double)acc;
                    }
                    dtb += d_dt;
                }
            }
            /* t' = t + dt */
            dtb += tb;
            /* dtp' = min(dt/q, dtmax) */
            if (r->dtmax_clamped) { t1b += dtpb; t0b -= dtpb; }
            else if (cfg->track_ctrl) { dtb += dtpb / (double)r->q; qb += -dtpb * (double)dt / ((double)r->q * (double)r->q); }
            /* qold' = max(EEst, qoldinit) */
            if (r->eest > QOLDINIT) eb += qoldb;
            qoldb_in = 0;
        } else {
            memset(unb, 0, sizeof(real) * N);
            memset(upb, 0, sizeof(real) * N);
            for (int j = 0; j < 7; ++j) memset(kb[j], 0, sizeof(real) * N);
            /* dtp' = dt / m (dtmax clamp after a reject cannot trigger: dt shrinks) */
            dtb += dtpb / (double)r->rej_m;
            if (r->rej_m_is_q11) q11b += -dtpb * (double)dt / ((double)r->rej_m * (double)r->rej_m) / (double)GAMMA;
            qoldb_in = qoldb;
        }
        /* q = clamp(q11 / qold^beta2 / gamma) */
        if (!r->q_clamped && !r->eest_zero) {
            double qo = pow((double)r->qold_in, (double)BETA2);
            q11b += qb / (qo * (double)GAMMA);
            qoldb_in += -(double)BETA2 * qb * (double)r->q / (double)r->qold_in;
        }
        /* q11 = EEst^beta1 */
        if (!r->eest_zero && r->eest > 0) eb += q11b * (double)BETA1 * (double)r->q11 / (double)r->eest;
        /* EEst = sqrt(sum r^2 / N); r = utilde/sk */
        {
            real bt[7];
            for (int j = 0; j < 7; ++j) bt[j] = (real)tabBT(cfg->solver)[j];
            double coef = (r->eest > 0) ? eb / ((double)N * (double)r->eest) : 0.0;
            double d_dt = 0;
            const size_t nch_ = (N + ORC_CHUNK - 1) / ORC_CHUNK;
            double* part_ = (double*)malloc(sizeof(double) * nch_);
#pragma omp parallel for schedule(static)
            for (size_t ch = 0; ch < nch_; ++ch) {
                const size_t i1_ = (ch + 1) * ORC_CHUNK < N ? (ch + 1) * ORC_CHUNK : N;
                double cs = 0;
                for (size_t i = ch * ORC_CHUNK; i < i1_; ++i) {
                    real acc = 0;
                    for (int j = 0; j < 7; ++j) acc += bt[j] * r->k[j][i];
                    real ut = dt * acc;
                    real au = rfabs(r->uprev[i]), an = rfabs(r->unew[i]);
                    int use_new = !(au > an);
                    real sk = cfg->abstol + (use_new ? an : au) * cfg->reltol;
                    real rr = ut / sk;
                    real rb = (real)(coef * (double)rr);
                    real utbv = rb / sk;
                    real skb = -rb * rr / sk;
                    utb[i] = utbv;
                    if (use_new) unb[i] += skb * cfg->reltol * (r->unew[i] > 0 ? 1 : (r->unew[i] < 0 ? -1 : 0));
                    else upb[i] += skb * cfg->reltol * (r->uprev[i] > 0 ? 1 : (r->uprev[i] < 0 ? -1 : 0));
                    for (int j = 0; j < 7; ++j) kb[j][i] += dt * bt[j] * utbv;
                    cs += (double)utbv * (double)acc;
                }
                part_[ch] = cs;
            }
            for (size_t ch = 0; ch < nch_; ++ch) d_dt += part_[ch];
            free(part_);
            dtb += d_dt;
        }
        /* stages 7..2 */
        for (int s = 6; s >= 1; --s) {
            /* stage input g_s+1 */
            real as[6];
            for (int j = 0; j < s; ++j) as[j] = (real)tabA(cfg->solver)[s][j];
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < N; ++i) {
                real acc = 0;
                for (int j = 0; j < s; ++j) acc += as[j] * r->k[j][i];
                gtmp[i] = r->uprev[i] + dt * acc;
            }
            const real* gin = (s == 6) ? r->unew : gtmp;
            real tst = t + (real)tabC(cfg->solver)[s] * dt;
            real taub = orc_f_backward(a, p, gin, r->acts[s], B, tst, kb[s], gb, pbar);
            tb_in += (double)taub;
            dtb += tabC(cfg->solver)[s] * (double)taub;
            if (s == 6) {
                /* k7 = f(unew): gbar adds to unew-bar, then unew = uprev + dt sum a7j kj */
#pragma omp parallel for schedule(static)
                for (size_t i = 0; i < N; ++i) unb[i] += gb[i];
                double d_dt = 0;
                const size_t nch_ = (N + ORC_CHUNK - 1) / ORC_CHUNK;
                double* part_ = (double*)malloc(sizeof(double) * nch_);
#pragma omp parallel for schedule(static)
                for (size_t ch = 0; ch < nch_; ++ch) {
                    const size_t i1_ = (ch + 1) * ORC_CHUNK < N ? (ch + 1) * ORC_CHUNK : N;
                    double cs = 0;
                    for (size_t i = ch * ORC_CHUNK; i < i1_; ++i) {
                        real acc = 0;
                        for (int j = 0; j < 6; ++j) { kb[j][i] += dt * as[j] * unb[i]; acc += as[j] * r->k[j][i]; }
                        upb[i] += unb[i];
                        cs += (double)unb[i] * (double)acc;
                    }
                    part_[ch] = cs;
                }
                for (size_t ch = 0; ch < nch_; ++ch) d_dt += part_[ch];
                free(part_);
                dtb += d_dt;
            } else {
                double d_dt = 0;
                const size_t nch_ = (N + ORC_CHUNK - 1) / ORC_CHUNK;
                double* part_ = (double*)malloc(sizeof(double) * nch_);
#pragma omp parallel for schedule(static)
                for (size_t ch = 0; ch < nch_; ++ch) {
                    const size_t i1_ = (ch + 1) * ORC_CHUNK < N ? (ch + 1) * ORC_CHUNK : N;
                    double cs = 0;
                    for (size_t i = ch * ORC_CHUNK; i < i1_; ++i) {
                        real acc = 0;
                        for (int j = 0; j < s; ++j) { kb[j][i] += dt * as[j] * gb[i]; acc += as[j] * r->k[j][i]; }
                        upb[i] += gb[i];
                        cs += (double)gb[i] * (double)acc;
                    }
                    part_[ch] = cs;
                }
                for (size_t ch = 0; ch < nch_; ++ch) d_dt += part_[ch];
                free(part_);
                dtb += d_dt;
            }
        }
        /* state cotangents out */
        if (r->accepted) {
            memcpy(U, upb, sizeof(real) * N);
            memcpy(K1, kb[0], sizeof(real) * N);
        } else {
            for (size_t i = 0; i < N; ++i) { U[i] += upb[i]; K1[i] += kb[0][i]; }
        }
        /* dt = min(dtp, t1 - t) */
        if (r->clamped) { t1b += dtb; tb_in -= dtb; dtpb = 0; }
        else dtpb = dtb;
        tb = tb_in;
        qoldb = qoldb_in;
    }
    /* k1 = f(u0, t0) (fsalfirst) */
    real* f0b = (real*)calloc(N, sizeof(real));
    memcpy(f0b, K1, sizeof(real) * N);
    real* u0b = ralloc(N);
    memcpy(u0b, U, sizeof(real) * N);
    if (nsave && h->save_t0)
        for (int c = 0; c < B; ++c)
            for (int i = 0; i < D; ++i) u0b[(size_t)c * D + i] += ubar[((size_t)c * nsave + 0) * D + i];
    /* initial dt */
    if (cfg->track_initdt && dtpb != 0) {
        double dtb = dtpb, dt0b = 0, d0b = 0, d1b = 0, d2b = 0;
        real dt0 = h->id_dt0;
        if (h->id_sel == 2) { t1b += dtb; t0b -= dtb; }
        else if (h->id_sel == 0) dt0b += 100.0 * dtb;
        else if (!h->id_dt1_const) {
            double m = h->id_max_is_d2 ? h->id_d2 : h->id_d1;
            double mb = dtb * (-0.2) * (double)h->id_dt1 / m;
            if (h->id_max_is_d2) d2b += mb; else d1b += mb;
        } else {
            /* dt1 = max(1e-6, 1e-3 dt0) */
            if (dt0 * R(1e-3) > R(1e-6)) dt0b += 1e-3 * dtb;
        }
        real* sk = ralloc(N);
        real* skb = (real*)calloc(N, sizeof(real));
        for (size_t i = 0; i < N; ++i) sk[i] = cfg->abstol + rfabs(h->u0[i]) * cfg->reltol;
        /* d2 = rms((f1-f0)/sk)/dt0 */
        real* f1b = (real*)calloc(N, sizeof(real));
        if (d2b != 0) {
            double n2 = (double)h->id_d2 * (double)dt0;
            double n2b = d2b / (double)dt0;
            dt0b += -d2b * (double)h->id_d2 / (double)dt0;
            for (size_t i = 0; i < N; ++i) {
                real w = (h->f1[i] - h->f0[i]) / sk[i];
                real wb = (real)(n2 > 0 ? n2b * (double)w / ((double)N * n2) : 0.0);
                f1b[i] += wb / sk[i];
                f0b[i] -= wb / sk[i];
                skb[i] += -wb * w / sk[i];
            }
            real taub = orc_f_backward(a, p, h->u1, h->acts1, B, h->t0 + dt0, f1b, gb, pbar);
            t0b += (double)taub;
            dt0b += (double)taub;
            for (size_t i = 0; i < N; ++i) { u0b[i] += gb[i]; f0b[i] += dt0 * gb[i]; }
            dt0b += dotp(gb, h->f0, N);
        }
        if (h->id_dt0_clamped) { t1b += dt0b; t0b -= dt0b; }
        else if (!h->id_dt0_const) {
            d0b += dt0b / (100.0 * (double)h->id_d1);
            d1b += -dt0b * (double)dt0 / (double)h->id_d1;
        }
        for (size_t i = 0; i < N; ++i) {
            if (d1b != 0) {
                real v = h->f0[i] / sk[i];
                real vb = (real)(d1b * (double)v / ((double)N * (double)h->id_d1));
                f0b[i] += vb / sk[i];
                skb[i] += -vb * v / sk[i];
            }
            if (d0b != 0) {
                real z = h->u0[i] / sk[i];
                real zb = (real)(d0b * (double)z / ((double)N * (double)h->id_d0));
                u0b[i] += zb / sk[i];
                skb[i] += -zb * z / sk[i];
            }
            u0b[i] += skb[i] * cfg->reltol * (h->u0[i] > 0 ? 1 : (h->u0[i] < 0 ? -1 : 0));
        }
        free(sk); free(skb); free(f1b);
    }
    {
        real taub = orc_f_backward(a, p, h->u0, h->acts0, B, h->t0, f0b, gb, pbar);
        t0b += (double)taub;
        for (size_t i = 0; i < N; ++i) u0b[i] += gb[i];
    }
    t0b += tb;
    memcpy(xbar, u0b, sizeof(real) * N);
    if (tspanbar) { tspanbar[0] = (real)t0b; tspanbar[1] = (real)t1b; }
    free(f0b); free(u0b); free(U); free(K1);
    for (int j = 0; j < 7; ++j) free(kb[j]);
    free(unb); free(upb); free(gb); free(utb); free(gtmp);
    return 0;
}
