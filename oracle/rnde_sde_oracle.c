/*
 * rnde_sde_oracle.c -- CPU ORACLE for TrackedNeuralDSDE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See rnde_sde_oracle.h.
 * PARITY UNPINNED versus the Julia reference (no Julia, no golden vectors; see the header).
 *
 * What each block follows (all [RECALL] of the pinned upstream packages unless a reference line is cited):
 *   drift / diffusion closures   reference src/models/neural_sde.jl:45-52 (p split at n.len, :17), experiments/mnist_nsde.jl:72-76
 *   tableaux                     StochasticDiffEq 6.30.1 src/tableaus.jl constructSOSRI / constructSOSRI2 / constructSRIW1
 *   one step + error estimate    StochasticDiffEq src/perform_step/sri.jl (FourStageSRIConstantCache), SURVEY.md B.7;
 *                                DiffEqBase calculate_residuals(E1, E2, uprev, u, abstol, reltol, delta, t)
 *   initial dt                   StochasticDiffEq src/initdt.jl sde_determine_initdt (out of place, diagonal noise)
 *   controller / loop            StochasticDiffEq src/integrators/integrator_utils.jl loopheader!/loopfooter!, src/alg_utils.jl defaults
 *   rejection sampling w/ memory DiffEqNoiseProcess accept_step!/reject_step! with RSWM(adaptivealg = :RSwM3)
 *                                (Rackauckas & Nie 2017, Algorithm RSwM3)
 *   saving callback              DiffEqCallbacks via reference neural_sde.jl:95-96, :127-128
 *   stiffness estimate           [RECALL] StochasticDiffEq src/perform_step/sri.jl, the block behind the error estimate of the four-stage SRI
 *                                step: `if alg isa StochasticCompositeAlgorithm && alg.algs[1] isa SOSRI2: eigen_est =
 *                                internalnorm(k4 - k3) / internalnorm(H0[4] - H0[3])` -- SOSRI2's last two drift stages share the time t + dt
 *                                (c0[3] = c0[4] = 1), so the quotient estimates |J_drift|.  Recorded by the experiment's callback as
 *                                |eigen_est| / alg_stability_size(SOSRI2()) = / 10.6 (reference experiments/mnist_nsde.jl:51-61, the shipped
 *                                configs/mnist_nsde.yml:6).  AutoSOSRI2(SOSRI2()) switches between two copies of the SAME method: the
 *                                trajectory is SOSRI2's (as AutoTsit5(Tsit5()) in the ODE oracle).  reg_kind = 2; the constant is cfg.stability_size.
 *   NFE counters                 reference neural_sde.jl:46,:50 (counted in the closures: the two probing evaluations of the
 *                                initial-step rule are included), returned at :109-113
 *   reverse pass                 what Tracker computes for sensealg = SensitivityADPassThrough() (neural_sde.jl:104,:136)
 */
#include "rnde_sde_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef RNDE_F64
#define R(x) x
#define rsqrt_ sqrt
#define rfabs fabs
#define rpow pow
#define rlog10 log10
#else
#define R(x) x##f
#define rsqrt_ sqrtf
#define rfabs fabsf
#define rpow powf
#define rlog10 log10f
#endif

/* ---------------- tableaux ---------------- */
static void tri(double* M, double a21, double a31, double a32, double a41, double a42, double a43) {
    memset(M, 0, 16 * sizeof(double));
    M[4] = a21; M[8] = a31; M[9] = a32; M[12] = a41; M[13] = a42; M[14] = a43;
}
int orc_sri_tableau_get(int id, orc_sri_tableau* T) {
    memset(T, 0, sizeof(*T));
    T->order = 1.5;
    if (id == 0) { /* SOSRI */
        tri(T->A0, -0.04199224421316468, 2.842612915017106, -2.0527723684000727, 4.338237071435815, -2.8895936137439793, 2.3017575594644466);
        tri(T->A1, 0.26204282091330466, 0.20903646383505375, -0.1502377115150361, 0.05836595312746999, 0.6149440396332373, 0.08535117634046772);
        tri(T->B0, -0.21641093549612528, 1.5336352863679572, 0.26066223492647056, -1.0536037558179159, 1.7015284721089472, -0.20725685784180017);
        tri(T->B1, -0.5119011827621657, 2.67767339866713, -4.9395031322250995, 0.15580956238299215, 3.2361551006624674, -1.4223118283355949);
        const double al[4] = {1.140099274172029, -0.6401334255743456, 0.4736296532772559, 0.026404498125060714};
        const double b1[4] = {-1.8453464565104432, 2.688764531100726, -0.2523866501071323, 0.40896857551684956};
        const double b2[4] = {0.4969658141589478, -0.5771202869753592, -0.12919702470322217, 0.2093514975196336};
        const double b3[4] = {2.8453464565104425, -2.688764531100725, 0.2523866501071322, -0.40896857551684945};
        const double b4[4] = {0.11522663875443433, -0.57877086147738, 0.2857851028163886, 0.17775911990655704};
        const double c0[4] = {0.0, -0.04199224421316468, 0.7898405466170333, 3.7504010171562823};
        const double c1[4] = {0.0, 0.26204282091330466, 0.05879875232001766, 0.758661169101175};
        memcpy(T->alpha, al, sizeof al); memcpy(T->beta1, b1, sizeof b1); memcpy(T->beta2, b2, sizeof b2);
        memcpy(T->beta3, b3, sizeof b3); memcpy(T->beta4, b4, sizeof b4); memcpy(T->c0, c0, sizeof c0); memcpy(T->c1, c1, sizeof c1);
        T->delta = 1.0;
        return 0;
    }
    if (id == 1) { /* SRIW1 (Roessler 2010) */
        tri(T->A0, 0.75, 0, 0, 0, 0, 0);
        tri(T->A1, 0.25, 1, 0, 0, 0, 0.25);
        tri(T->B0, 1.5, 0, 0, 0, 0, 0);
        tri(T->B1, 0.5, -1, 0, -5, 3, 0.5);
        const double al[4] = {1.0 / 3, 2.0 / 3, 0, 0};
        const double b1[4] = {-1, 4.0 / 3, 2.0 / 3, 0};
        const double b2[4] = {-1, 4.0 / 3, -1.0 / 3, 0};
        const double b3[4] = {2, -4.0 / 3, -2.0 / 3, 0};
        const double b4[4] = {-2, 5.0 / 3, -2.0 / 3, 1};
        const double c0[4] = {0, 0.75, 0, 0};
        const double c1[4] = {0, 0.25, 1, 0.25};
        memcpy(T->alpha, al, sizeof al); memcpy(T->beta1, b1, sizeof b1); memcpy(T->beta2, b2, sizeof b2);
        memcpy(T->beta3, b3, sizeof b3); memcpy(T->beta4, b4, sizeof b4); memcpy(T->c0, c0, sizeof c0); memcpy(T->c1, c1, sizeof c1);
        T->delta = 1.0 / 6.0;
        return 0;
    }
    if (id == 2) { /* SOSRI2 */
        tri(T->A0, 0.13804532298278663, 0.5818361298250374, 0.4181638701749618, 0.4670018408674211, 0.8046204792187386, -0.27162232008616016);
        tri(T->A1, 0.45605532163856893, 0.7555807846451692, 0.24441921535482677, 0.6981181143266059, 0.3453277086024727, -0.04344582292908241);
        tri(T->B0, 0.08852381537667678, 1.0317752458971061, 0.4563552922077882, 1.73078280444124, -0.46089678470929774, -0.9637509618944188);
        tri(T->B1, 0.6753186815412179, -0.07452812525785148, -0.49783736486149366, -0.5591906709928903, 0.022696571806569924, -0.8984927888368557);
        const double al[4] = {-0.15036858140642623, 0.7545275856696072, 0.686995463807979, -0.2911544680711602};
        const double b1[4] = {-0.45315689727309133, 0.8330937231303951, 0.3792843195533544, 0.24077885458934192};
        const double b2[4] = {-0.4994383733810986, 0.9181786186154077, -0.25613778661003145, -0.16260245862427797};
        const double b3[4] = {1.4531568972730915, -0.8330937231303933, -0.3792843195533583, -0.24077885458934023};
        const double b4[4] = {-0.4976090683622265, 0.9148155835648892, -1.4102107084476505, 0.9930042001464879};
        const double c0[4] = {0.0, 0.13804532298278663, 0.9999999999999992, 0.9999999999999994};
        const double c1[4] = {0.0, 0.45605532163856893, 0.999999999999996, 0.9999999999999962};
        memcpy(T->alpha, al, sizeof al); memcpy(T->beta1, b1, sizeof b1); memcpy(T->beta2, b2, sizeof b2);
        memcpy(T->beta3, b3, sizeof b3); memcpy(T->beta4, b4, sizeof b4); memcpy(T->c0, c0, sizeof c0); memcpy(T->c1, c1, sizeof c1);
        T->delta = 1.0;
        return 0;
    }
    return -1;
}

/* ---------------- handle ---------------- */
typedef struct { real L; real* w; real* z; } stack_item;
typedef struct { stack_item* it; int n, cap; } nstack;

typedef struct {
    real t, dt, eest;
    real n1, n2;              /* rms(k4 - k3), rms(H0_4 - H0_3): eigen_est = n1 / n2 (reg_kind 2) */
    int sv_index;
    int sv_first, nsv_pts;    /* saveat points filled from this step (linear interpolation between uprev and u) */
    real *uprev, *u;          /* uprev borrowed (u0 or the previous record's u), u owned */
    real *dW, *dZ;            /* owned copies of the increments the attempt used */
    real *k[4], *g[4];        /* owned */
    real *H0[4], *H1[4];      /* stage inputs (index 0 = uprev, borrowed), 1..3 owned */
    real *actf[4], *actg[4];  /* activations of the 8 evaluations */
} sde_rec;

typedef struct {
    orc_sde_config cfg;
    orc_sri_tableau T;
    int D, P, Pf, Pg, B, arows_f, arows_g;
    real beta1, beta2, gamma, qmin, qmax, qoldinit, delta;
    /* tape of ACCEPTED attempts (rejected ones carry no gradient: the controller strips tracking) */
    int n_acc;
    sde_rec* rec;
    real *u0, *p;
    real *wtot, *ztot;
    int have_tape, n_saveval;
    int n_replay; real* replay_dt; int* replay_acc;   /* orc_sde_set_replay */
    int last_is_attempt; real last_nrm[2];                                 /* orc_sde_attempt: rms(k4 - k3), rms(H0_4 - H0_3) of the last call */
    int nsave; real* saveat; int save_t0;             /* orc_sde_set_saveat: the {R,true} call methods (neural_sde.jl:44-61,:84-113) */
} sde_handle;

static real* ralloc(size_t n) { return (real*)malloc(sizeof(real) * n); }
static real* rdup(const real* a, size_t n) { real* r = ralloc(n); memcpy(r, a, sizeof(real) * n); return r; }

int orc_sde_param_count(const orc_sde_config* cfg, int* len_drift) {
    int a = orc_param_count(&cfg->drift), b = orc_param_count(&cfg->diffusion);
    if (len_drift) *len_drift = a;
    return a + b;
}

void* orc_sde_create(const orc_sde_config* cfg) {
    sde_handle* h = (sde_handle*)calloc(1, sizeof(sde_handle));
    h->cfg = *cfg;
    if (orc_sri_tableau_get(cfg->tableau, &h->T) != 0) { free(h); return NULL; }
    h->D = cfg->drift.dims[0];
    h->Pf = orc_param_count(&cfg->drift); h->Pg = orc_param_count(&cfg->diffusion); h->P = h->Pf + h->Pg;
    h->arows_f = orc_act_rows_total(&cfg->drift); h->arows_g = orc_act_rows_total(&cfg->diffusion);
    const double order = h->T.order;
    h->beta2 = cfg->beta2 != 0 ? cfg->beta2 : (real)(2.0 / (5.0 * order));
    h->beta1 = cfg->beta1 != 0 ? cfg->beta1 : (real)(7.0 / (10.0 * order));
    h->gamma = cfg->gamma != 0 ? cfg->gamma : (real)0.9;
    h->qmin = cfg->qmin != 0 ? cfg->qmin : (real)0.2;
    h->qmax = cfg->qmax != 0 ? cfg->qmax : (real)1.125;
    h->qoldinit = cfg->qoldinit != 0 ? cfg->qoldinit : (real)1e-4;
    h->delta = cfg->delta != 0 ? cfg->delta : (real)h->T.delta;
    h->rec = (sde_rec*)calloc((size_t)cfg->max_attempts + 1, sizeof(sde_rec));
    return h;
}
static void free_rec(sde_rec* r) {
    free(r->u); free(r->dW); free(r->dZ);
    for (int j = 0; j < 4; ++j) { free(r->k[j]); free(r->g[j]); free(r->actf[j]); free(r->actg[j]); if (j) { free(r->H0[j]); free(r->H1[j]); } }
    memset(r, 0, sizeof(*r));
}
static void free_tape(sde_handle* h) {
    for (int n = 0; n < h->n_acc; ++n) free_rec(&h->rec[n]);
    h->n_acc = 0;
    free(h->u0); free(h->p); free(h->wtot); free(h->ztot);
    h->u0 = h->p = h->wtot = h->ztot = NULL;
    h->have_tape = 0;
}
/* saveat (the {R,true} methods): following forwards return the state at every time of `saveat` (increasing, inside [t0, t1]) as a
 * (D, T, B) column-major array; points inside a step come from the SDE solution's LINEAR interpolant (StochasticDiffEq has no
 * higher-order dense output), a point equal to t0 is u0 (save_start).  n = 0 switches back to the end state. */
void orc_sde_set_saveat(void* hh, const real* saveat, int n) {
    sde_handle* h = (sde_handle*)hh;
    free(h->saveat);
    h->saveat = NULL; h->nsave = 0;
    if (n <= 0) return;
    h->saveat = rdup(saveat, n);
    h->nsave = n;
}
void orc_sde_set_replay(void* hh, const real* dt, const int* acc, int n) {
    sde_handle* h = (sde_handle*)hh;
    free(h->replay_dt); free(h->replay_acc);
    h->replay_dt = NULL; h->replay_acc = NULL; h->n_replay = 0;
    if (n <= 0) return;
    h->replay_dt = rdup(dt, n);
    h->replay_acc = (int*)malloc(sizeof(int) * n);
    memcpy(h->replay_acc, acc, sizeof(int) * n);
    h->n_replay = n;
}
void orc_sde_destroy(void* hh) {
    sde_handle* h = (sde_handle*)hh;
    if (!h) return;
    free_tape(h);
    free(h->replay_dt); free(h->replay_acc); free(h->saveat);
    free(h->rec);
    free(h);
}

/* ---------------- one attempt ---------------- */
/* rec receives owned arrays when keep != 0; otherwise kg (8 arrays) and unew are caller buffers */
static real sde_attempt(const sde_handle* h, const real* p, const real* uprev, int B, real dt, const real* dW, const real* dZ,
                        real* k[4], real* g[4], real* H0[4], real* H1[4], real* actf[4], real* actg[4], real* unew, real* nrm) {
    const orc_sri_tableau* T = &h->T;
    const size_t N = (size_t)h->D * B;
    const real sqdt = rsqrt_(rfabs(dt));
    const real sqrt3 = rsqrt_((real)3);
    const real* pf = p;
    const real* pg = p + h->Pf;
    real* chi2 = ralloc(N);
    for (size_t i = 0; i < N; ++i) chi2[i] = (dW[i] + dZ[i] / sqrt3) / 2;
    for (int s = 0; s < 4; ++s) {
        const real* h0 = uprev;
        const real* h1 = uprev;
        if (s > 0) {
            for (size_t i = 0; i < N; ++i) {
                real a0 = 0, b0 = 0, a1 = 0, b1 = 0;
                for (int j = 0; j < s; ++j) {
                    a0 += (real)T->A0[4 * s + j] * k[j][i]; b0 += (real)T->B0[4 * s + j] * g[j][i];
                    a1 += (real)T->A1[4 * s + j] * k[j][i]; b1 += (real)T->B1[4 * s + j] * g[j][i];
                }
                H0[s][i] = uprev[i] + dt * a0 + chi2[i] * b0;
                H1[s][i] = uprev[i] + dt * a1 + sqdt * b1;
            }
            h0 = H0[s]; h1 = H1[s];
        }
        orc_f_forward(&h->cfg.drift, pf, h0, B, 0, k[s], actf ? actf[s] : NULL);
        orc_f_forward(&h->cfg.diffusion, pg, h1, B, 0, g[s], actg ? actg[s] : NULL);
    }
    double ssum = 0;
    for (size_t i = 0; i < N; ++i) {
        const real w = dW[i];
        const real chi1 = (w * w - rfabs(dt)) / (2 * sqdt);
        const real chi3 = (w * w * w - 3 * w * dt) / (6 * dt);
        real sa = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, sk_ = 0;
        for (int j = 0; j < 4; ++j) {
            sa += (real)T->alpha[j] * k[j][i]; sk_ += k[j][i];
            s1 += (real)T->beta1[j] * g[j][i]; s2 += (real)T->beta2[j] * g[j][i];
            s3 += (real)T->beta3[j] * g[j][i]; s4 += (real)T->beta4[j] * g[j][i];
        }
        const real E2 = chi2[i] * s3 + chi3 * s4;
        const real u = uprev[i] + dt * sa + E2 + w * s1 + chi1 * s2;
        unew[i] = u;
        const real E1 = dt * sk_;
        const real au = rfabs(uprev[i]), an = rfabs(u);
        const real sc = h->cfg.abstol + (au > an ? au : an) * h->cfg.reltol;
        const real r = (h->delta * E1 + E2) / sc;
        ssum += (double)(r * r);
    }
    free(chi2);
    if (nrm) {   /* the two norms of the stiffness estimate (DiffEqBase ODE_DEFAULT_NORM: rms), from the LAST TWO drift stages */
        double s1 = 0, s2 = 0;
        for (size_t i = 0; i < N; ++i) {
            const real v1 = k[3][i] - k[2][i], v2 = H0[3][i] - H0[2][i];
            s1 += (double)(v1 * v1); s2 += (double)(v2 * v2);
        }
        nrm[0] = (real)sqrt(s1 / (double)N); nrm[1] = (real)sqrt(s2 / (double)N);
    }
    return (real)sqrt(ssum / (double)N);
}
static real sde_stab(const sde_handle* h) { return h->cfg.stability_size != 0 ? h->cfg.stability_size : (real)10.6; }
static real sde_cb_value(const sde_handle* h, real eest, real dt, real eigen) {   /* the experiment's save_func (mnist_nsde.jl:48, :53-58) */
    if (h->cfg.reg_kind == 1) return eest * dt;
    if (h->cfg.reg_kind == 2) { const real a = rfabs(eigen); return (a == 0 || isnan(a)) ? 0 : a / sde_stab(h); }
    return 0;
}

void orc_sde_attempt(void* hh, const real* p, const real* uprev, int B, real dt, const real* dW, const real* dZ,
                     real* kg_out, real* unew, real* eest) {
    sde_handle* h = (sde_handle*)hh;
    const size_t N = (size_t)h->D * B;
    real *k[4], *g[4], *H0[4], *H1[4];
    for (int j = 0; j < 4; ++j) { k[j] = kg_out + (size_t)j * N; g[j] = kg_out + (size_t)(4 + j) * N; H0[j] = j ? ralloc(N) : NULL; H1[j] = j ? ralloc(N) : NULL; }
    *eest = sde_attempt(h, p, uprev, B, dt, dW, dZ, k, g, H0, H1, NULL, NULL, unew, h->last_nrm);
    h->last_is_attempt = 1;
    for (int j = 1; j < 4; ++j) { free(H0[j]); free(H1[j]); }
}

/* ---------------- noise process with rejection-sampling memory (RSwM3) ---------------- */
static void st_push(nstack* s, real L, real* w, real* z) {
    if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 16; s->it = (stack_item*)realloc(s->it, sizeof(stack_item) * s->cap); }
    s->it[s->n].L = L; s->it[s->n].w = w; s->it[s->n].z = z; ++s->n;
}
static void st_clear(nstack* s) { for (int i = 0; i < s->n; ++i) { free(s->it[i].w); free(s->it[i].z); } s->n = 0; }

typedef struct {
    size_t N;
    real dt; real *dW, *dZ;
    nstack S1, S2;           /* S1: the known future of the path; S2: the pieces the current step is made of (oldest first) */
    const real* pool; int n_pool, next;
    real discard;
} noise_t;

static const real* noise_draw(noise_t* W) { return W->next < W->n_pool ? W->pool + (size_t)(W->next++) * 2 * W->N : NULL; }

/* first step: dW, dZ ~ N(0, dt) */
static int noise_init(noise_t* W, real dt) {
    const real* xi = noise_draw(W);
    if (!xi) return 4;
    const real s = rsqrt_(rfabs(dt));
    for (size_t i = 0; i < W->N; ++i) { W->dW[i] = s * xi[i]; W->dZ[i] = s * xi[W->N + i]; }
    W->dt = dt;
    st_push(&W->S2, dt, rdup(W->dW, W->N), rdup(W->dZ, W->N));
    return 0;
}
/* after an accepted step: increments for the next step of size dt, first from the stack of known future pieces */
static int noise_accept(noise_t* W, real dt) {
    const size_t N = W->N;
    st_clear(&W->S2);
    real dttmp = 0;
    memset(W->dW, 0, sizeof(real) * N); memset(W->dZ, 0, sizeof(real) * N);
    int bridged = 0;
    while (W->S1.n > 0) {
        stack_item it = W->S1.it[--W->S1.n];
        const real qtmp = (dt - dttmp) / it.L;
        if (qtmp > 1) {            /* the whole piece lies inside the step */
            dttmp += it.L;
            for (size_t i = 0; i < N; ++i) { W->dW[i] += it.w[i]; W->dZ[i] += it.z[i]; }
            st_push(&W->S2, it.L, it.w, it.z);
        } else {                   /* the step ends inside this piece: Brownian bridge at fraction qtmp */
            const real* xi = noise_draw(W);
            if (!xi) { free(it.w); free(it.z); return 4; }
            const real sd = rsqrt_((1 - qtmp) * qtmp * it.L);
            real* bw = ralloc(N); real* bz = ralloc(N);
            for (size_t i = 0; i < N; ++i) {
                bw[i] = qtmp * it.w[i] + sd * xi[i]; bz[i] = qtmp * it.z[i] + sd * xi[N + i];
                W->dW[i] += bw[i]; W->dZ[i] += bz[i];
                it.w[i] -= bw[i]; it.z[i] -= bz[i];
            }
            const real rest = (1 - qtmp) * it.L;
            if (rest > W->discard) st_push(&W->S1, rest, it.w, it.z); else { free(it.w); free(it.z); }
            if (qtmp * it.L > W->discard) st_push(&W->S2, qtmp * it.L, bw, bz); else { free(bw); free(bz); }
            bridged = 1;
            break;
        }
    }
    if (!bridged) {
        const real dtleft = dt - dttmp;
        if (dtleft > 0) {          /* stack emptied: fresh noise for the rest */
            const real* xi = noise_draw(W);
            if (!xi) return 4;
            const real s = rsqrt_(dtleft);
            real* fw = ralloc(N); real* fz = ralloc(N);
            for (size_t i = 0; i < N; ++i) { fw[i] = s * xi[i]; fz[i] = s * xi[N + i]; W->dW[i] += fw[i]; W->dZ[i] += fz[i]; }
            st_push(&W->S2, dtleft, fw, fz);
        }
    }
    W->dt = dt;
    return 0;
}
/* after a rejected step: shrink the step to dtnew, keeping everything already drawn */
static int noise_reject(noise_t* W, real dtnew) {
    const size_t N = W->N;
    real dttmp = 0;
    real* tw = (real*)calloc(N, sizeof(real)); real* tz = (real*)calloc(N, sizeof(real));
    while (W->S2.n > 0) {          /* whole pieces of the tail go back to the future stack (last piece first) */
        stack_item it = W->S2.it[W->S2.n - 1];
        if (W->dt - dttmp - it.L < dtnew) break;     /* this piece straddles (or precedes) the new end */
        --W->S2.n;
        dttmp += it.L;
        for (size_t i = 0; i < N; ++i) { tw[i] += it.w[i]; tz[i] += it.z[i]; }
        st_push(&W->S1, it.L, it.w, it.z);
    }
    const real dtK = W->dt - dttmp;
    const real qK = dtnew / dtK;
    const real* xi = noise_draw(W);
    if (!xi) { free(tw); free(tz); return 4; }
    const real sd = rsqrt_((1 - qK) * qK * dtK);
    real* rw = ralloc(N); real* rz = ralloc(N);
    for (size_t i = 0; i < N; ++i) {
        const real K2 = W->dW[i] - tw[i], K3 = W->dZ[i] - tz[i];
        const real bw = qK * K2 + sd * xi[i], bz = qK * K3 + sd * xi[N + i];
        rw[i] = K2 - bw; rz[i] = K3 - bz;
        W->dW[i] = bw; W->dZ[i] = bz;
    }
    free(tw); free(tz);
    const real cut = (1 - qK) * dtK;
    if (cut > W->discard) st_push(&W->S1, cut, rw, rz); else { free(rw); free(rz); }
    /* the current step is now ONE piece (the finer structure of [0, dtK] is forgotten, never reused) */
    st_clear(&W->S2);
    st_push(&W->S2, dtnew, rdup(W->dW, N), rdup(W->dZ, N));
    W->dt = dtnew;
    return 0;
}

/* ---------------- initial dt (sde_determine_initdt) ---------------- */
static real rms_ratio(const real* a, const real* sk, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) { real v = a[i] / sk[i]; s += (double)(v * v); }
    return (real)sqrt(s / (double)n);
}
static real sde_initdt(const sde_handle* h, const real* p, const real* u0, int B, real t0, real t1) {
    const size_t N = (size_t)h->D * B;
    const real dtmax = t1 - t0;
    real *sk = ralloc(N), *f0 = ralloc(N), *g0 = ralloc(N), *u1 = ralloc(N), *f1 = ralloc(N), *g1 = ralloc(N), *tmp = ralloc(N);
    for (size_t i = 0; i < N; ++i) sk[i] = h->cfg.abstol + rfabs(u0[i]) * h->cfg.reltol;
    const real d0 = rms_ratio(u0, sk, N);
    orc_f_forward(&h->cfg.drift, p, u0, B, 0, f0, NULL);
    orc_f_forward(&h->cfg.diffusion, p + h->Pf, u0, B, 0, g0, NULL);
    for (size_t i = 0; i < N; ++i) { g0[i] *= 3; const real a = rfabs(f0[i] + g0[i]), b = rfabs(f0[i] - g0[i]); tmp[i] = a > b ? a : b; }
    const real d1 = rms_ratio(tmp, sk, N);
    real dt0 = (d0 < R(1e-5) || d1 < R(1e-5)) ? R(1e-6) : (d0 / d1) / R(100.0);
    if (dtmax < dt0) dt0 = dtmax;
    for (size_t i = 0; i < N; ++i) u1[i] = u0[i] + dt0 * f0[i];
    orc_f_forward(&h->cfg.drift, p, u1, B, 0, f1, NULL);
    orc_f_forward(&h->cfg.diffusion, p + h->Pf, u1, B, 0, g1, NULL);
    for (size_t i = 0; i < N; ++i) {
        g1[i] *= 3;
        const real da = rfabs(g0[i] - g1[i]), db = rfabs(g0[i] + g1[i]);
        const real dg = da > db ? da : db;
        const real a = rfabs(f1[i] - f0[i] + dg), b = rfabs(f1[i] - f0[i] - dg);
        tmp[i] = a > b ? a : b;
    }
    const real d2 = rms_ratio(tmp, sk, N) / dt0;
    const real m = d1 > d2 ? d1 : d2;
    real dt1;
    if (m <= R(1e-15)) { const real a1 = R(1e-6), a2 = dt0 * R(1e-3); dt1 = a1 > a2 ? a1 : a2; }
    else dt1 = (real)pow(10.0, (double)(-(R(2.0) + rlog10(m)) / (real)(h->T.order + 0.5)));
    real dt = R(100.0) * dt0;
    if (dt1 < dt) dt = dt1;
    if (dtmax < dt) dt = dtmax;
    free(sk); free(f0); free(g0); free(u1); free(f1); free(g1); free(tmp);
    return dt;
}

/* ---------------- forward solve ---------------- */
int orc_sde_forward(void* hh, const real* x, const real* p, int B, real t0, real t1, const real* noise, int n_pool,
                    real* u_out, long* nfe1, long* nfe2, real* saveval, int* nsaveval, real* steps_log, int* nattempts,
                    int* ndraws_out) {
    sde_handle* h = (sde_handle*)hh;
    const orc_sde_config* cfg = &h->cfg;
    free_tape(h);
    h->last_is_attempt = 0;
    const size_t N = (size_t)h->D * B;
    h->B = B;
    h->p = rdup(p, h->P);
    h->u0 = rdup(x, N);
    h->wtot = (real*)calloc(N, sizeof(real)); h->ztot = (real*)calloc(N, sizeof(real));
    long nf1 = 0, nf2 = 0;
    const real dtmax = t1 - t0;
    const real dtmin = (real)(sizeof(real) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
    real dt = sde_initdt(h, p, x, B, t0, t1);
    nf1 += 2; nf2 += 2;
    const int replay = h->n_replay;
    if (replay) dt = h->replay_dt[0];
    real t = t0, qold = h->qoldinit;
    int nsv = 0, ret = 0, n = 0;
    if (cfg->reg_kind && cfg->cb_save_start) saveval[nsv++] = sde_cb_value(h, 1, 0, 1);   /* EEst = 1, dt = 0, eigen_est = 1 at callback initialisation */
    noise_t W;
    memset(&W, 0, sizeof(W));
    W.N = N; W.dW = ralloc(N); W.dZ = ralloc(N); W.pool = noise; W.n_pool = n_pool; W.discard = (real)1e-15;
    if (t1 - t < dt) dt = t1 - t;
    ret = noise_init(&W, dt);
    const real* uprev = h->u0;
    const int nsave = h->nsave, Dd = h->D;
    int next_save = 0;
    h->save_t0 = 0;
    if (nsave && h->saveat[0] == t0) {
        for (int c = 0; c < B; ++c) memcpy(u_out + ((size_t)c * nsave) * Dd, x + (size_t)c * Dd, sizeof(real) * Dd);
        next_save = 1; h->save_t0 = 1;
    }
    while (ret == 0 && t < t1 && (!replay || n < replay)) {
        if (n >= cfg->max_attempts) { ret = 1; break; }
        if (!(dt > dtmin) || isnan(dt)) { ret = isnan(dt) ? 3 : 2; break; }
        sde_rec r;
        memset(&r, 0, sizeof(r));
        r.t = t; r.dt = dt; r.sv_index = -1; r.uprev = (real*)uprev;
        r.u = ralloc(N);
        for (int j = 0; j < 4; ++j) {
            r.k[j] = ralloc(N); r.g[j] = ralloc(N);
            r.actf[j] = ralloc((size_t)h->arows_f * B); r.actg[j] = ralloc((size_t)h->arows_g * B);
            r.H0[j] = j ? ralloc(N) : NULL; r.H1[j] = j ? ralloc(N) : NULL;
        }
        real nrm[2] = {0, 0};
        const real eest = sde_attempt(h, p, uprev, B, dt, W.dW, W.dZ, r.k, r.g, r.H0, r.H1, r.actf, r.actg, r.u, nrm);
        r.n1 = nrm[0]; r.n2 = nrm[1];
        nf1 += 4; nf2 += 4;
        ++n;
        r.eest = eest;
        if (!(eest == eest) || isinf(eest)) { free_rec(&r); ret = 3; break; }
        const real q11 = rpow(eest, h->beta1);
        real q = q11 / rpow(qold, h->beta2);
        { const real qg = q / h->gamma, lo = 1 / h->qmax, hi = 1 / h->qmin; q = qg < lo ? lo : (qg > hi ? hi : qg); }
        const int accepted = replay ? h->replay_acc[n - 1] : (eest <= 1);
        if (steps_log) { steps_log[4 * (n - 1)] = t; steps_log[4 * (n - 1) + 1] = dt; steps_log[4 * (n - 1) + 2] = eest; steps_log[4 * (n - 1) + 3] = (real)accepted; }
        if (accepted) {
            r.dW = rdup(W.dW, N); r.dZ = rdup(W.dZ, N);
            for (size_t i = 0; i < N; ++i) { h->wtot[i] += W.dW[i]; h->ztot[i] += W.dZ[i]; }
            if (cfg->reg_kind) { r.sv_index = nsv; saveval[nsv++] = sde_cb_value(h, eest, dt, r.n1 / r.n2); }
            r.sv_first = next_save; r.nsv_pts = 0;
            {
                const real tnew = t + dt;
                while (next_save < nsave && h->saveat[next_save] <= tnew) {
                    const real th = (h->saveat[next_save] == tnew) ? (real)1 : (h->saveat[next_save] - t) / dt;
                    for (int c = 0; c < B; ++c)
                        for (int i = 0; i < Dd; ++i) {
                            const size_t e = (size_t)c * Dd + i;
                            u_out[((size_t)c * nsave + next_save) * Dd + i] = th == 1 ? r.u[e] : (1 - th) * uprev[e] + th * r.u[e];
                        }
                    ++next_save; ++r.nsv_pts;
                }
            }
            h->rec[h->n_acc++] = r;
            uprev = r.u;
            t = t + dt;
            qold = eest > h->qoldinit ? eest : h->qoldinit;
            real dtn = dt / q;
            if (dtmax < dtn) dtn = dtmax;
            if (dtn < dtmin) dtn = dtmin;
            if (replay && n < replay) dtn = h->replay_dt[n];
            if (!(t < t1) || (replay && n >= replay)) break;
            if (t1 - t < dtn) dtn = t1 - t;
            ret = noise_accept(&W, dtn);
            dt = dtn;
        } else {
            free_rec(&r);
            real m = 1 / h->qmin;
            const real m2 = q11 / h->gamma;
            if (m2 < m) m = m2;
            real dtn = dt / m;
            if (dtmax < dtn) dtn = dtmax;
            if (replay && n < replay) dtn = h->replay_dt[n];
            if (replay && n >= replay) break;
            if (t1 - t < dtn) dtn = t1 - t;
            ret = noise_reject(&W, dtn);
            dt = dtn;
        }
    }
    if (!nsave) memcpy(u_out, uprev, sizeof(real) * N);
    *nfe1 = nf1; *nfe2 = nf2; *nsaveval = nsv; *nattempts = n;
    if (ndraws_out) *ndraws_out = W.next;
    h->n_saveval = nsv;
    h->have_tape = (ret == 0);
    st_clear(&W.S1); st_clear(&W.S2); free(W.S1.it); free(W.S2.it); free(W.dW); free(W.dZ);
    return ret;
}

void orc_sde_path_total(void* hh, real* w_total, real* z_total) {
    sde_handle* h = (sde_handle*)hh;
    const size_t N = (size_t)h->D * h->B;
    memcpy(w_total, h->wtot, sizeof(real) * N);
    memcpy(z_total, h->ztot, sizeof(real) * N);
}

/* ---------------- reverse pass ---------------- */
int orc_sde_backward(void* hh, const real* ubar, const real* svbar, real* xbar, real* pbar) {
    sde_handle* h = (sde_handle*)hh;
    if (!h->have_tape) return -1;
    const orc_sri_tableau* T = &h->T;
    const int B = h->B;
    const size_t N = (size_t)h->D * B;
    const real* pf = h->p;
    const real* pg = h->p + h->Pf;
    memset(pbar, 0, sizeof(real) * h->P);
    const int nsave = h->nsave, Dd = h->D;
    real* U = nsave ? (real*)calloc(N, sizeof(real)) : rdup(ubar, N);   /* cotangent of the state after the step being reversed */
    real *kb[4], *gb[4];
    for (int j = 0; j < 4; ++j) { kb[j] = ralloc(N); gb[j] = ralloc(N); }
    real* upb = ralloc(N);
    real* hb = ralloc(N);
    const real sqrt3 = rsqrt_((real)3);
    for (int n = h->n_acc - 1; n >= 0; --n) {
        const sde_rec* r = &h->rec[n];
        const real dt = r->dt, sqdt = rsqrt_(rfabs(dt));
        /* saveat points of this step: u(ts) = (1 - th) uprev + th u, th a constant of the reverse pass */
        real* svup = NULL;
        if (r->nsv_pts) {
            svup = (real*)calloc(N, sizeof(real));
            for (int idx = r->sv_first; idx < r->sv_first + r->nsv_pts; ++idx) {
                const real tnew = r->t + dt;
                const real th = (h->saveat[idx] == tnew) ? (real)1 : (h->saveat[idx] - r->t) / dt;
                for (int c = 0; c < B; ++c)
                    for (int i = 0; i < Dd; ++i) {
                        const size_t e = (size_t)c * Dd + i;
                        const real ub = ubar[((size_t)c * nsave + idx) * Dd + i];
                        U[e] += th * ub;
                        svup[e] += (1 - th) * ub;
                    }
            }
        }
        double eb = 0;   /* cotangent of EEst: saveval = EEst * dt, dt a constant here */
        if (r->sv_index >= 0 && svbar && h->cfg.reg_kind == 1) eb = (double)svbar[r->sv_index] * (double)dt;
        const double coef = (r->eest > 0) ? eb / ((double)N * (double)r->eest) : 0.0;
        for (size_t i = 0; i < N; ++i) {
            const real w = r->dW[i];
            const real chi1 = (w * w - rfabs(dt)) / (2 * sqdt);
            const real chi2 = (w + r->dZ[i] / sqrt3) / 2;
            const real chi3 = (w * w * w - 3 * w * dt) / (6 * dt);
            real sk_ = 0, s3 = 0, s4 = 0;
            for (int j = 0; j < 4; ++j) { sk_ += r->k[j][i]; s3 += (real)T->beta3[j] * r->g[j][i]; s4 += (real)T->beta4[j] * r->g[j][i]; }
            const real E2 = chi2 * s3 + chi3 * s4, E1 = dt * sk_;
            const real au = rfabs(r->uprev[i]), an = rfabs(r->u[i]);
            const int use_new = !(au > an);
            const real sc = h->cfg.abstol + (use_new ? an : au) * h->cfg.reltol;
            const real res = (h->delta * E1 + E2) / sc;
            const real rb = (real)(coef * (double)res);
            const real numb = rb / sc;                 /* cotangent of delta*E1 + E2 */
            const real scb = -rb * res / sc;
            real unb = U[i];
            real up = 0;
            if (use_new) unb += scb * h->cfg.reltol * (r->u[i] > 0 ? 1 : (r->u[i] < 0 ? -1 : 0));
            else up += scb * h->cfg.reltol * (r->uprev[i] > 0 ? 1 : (r->uprev[i] < 0 ? -1 : 0));
            /* u = uprev + dt sum alpha k + E2 + dW sum beta1 g + chi1 sum beta2 g */
            up += unb;
            const real e2b = unb + numb;
            for (int j = 0; j < 4; ++j) {
                kb[j][i] = dt * (real)T->alpha[j] * unb + dt * h->delta * numb;
                gb[j][i] = (w * (real)T->beta1[j] + chi1 * (real)T->beta2[j]) * unb + (chi2 * (real)T->beta3[j] + chi3 * (real)T->beta4[j]) * e2b;
            }
            upb[i] = up + (svup ? svup[i] : 0);
        }
        free(svup);
        /* stiffness estimate: value = |n1 / n2| / stab with n1 = rms(k4 - k3), n2 = rms(H0_4 - H0_3); zero / NaN estimates are recorded as the
         * constant 0 (mnist_nsde.jl:55-57).  d n1 / d v1_i = v1_i / (N n1), so k4bar += c1 v1, k3bar -= c1 v1 and the stage inputs get
         * H0_4bar += c2 v2, H0_3bar -= c2 v2 (added to the drift's input cotangent of stages 4 and 3 below). */
        double c1 = 0, c2 = 0;
        if (r->sv_index >= 0 && svbar && h->cfg.reg_kind == 2 && r->n1 > 0 && r->n2 > 0) {
            const double eigb = (double)svbar[r->sv_index] / (double)sde_stab(h);
            c1 = eigb / ((double)N * (double)r->n1 * (double)r->n2);
            c2 = -eigb * (double)r->n1 / ((double)N * (double)r->n2 * (double)r->n2 * (double)r->n2);
            for (size_t i = 0; i < N; ++i) {
                const real v1 = r->k[3][i] - r->k[2][i];
                kb[3][i] += (real)(c1 * (double)v1); kb[2][i] -= (real)(c1 * (double)v1);
            }
        }
        for (int s = 3; s >= 0; --s) {
            const real* h0 = s ? r->H0[s] : r->uprev;
            const real* h1 = s ? r->H1[s] : r->uprev;
            /* k_s = f(H0_s) */
            orc_f_backward(&h->cfg.drift, pf, h0, r->actf[s], B, 0, kb[s], hb, pbar);
            if (c2 != 0 && s >= 2)
                for (size_t i = 0; i < N; ++i) { const real v2 = r->H0[3][i] - r->H0[2][i]; hb[i] += (real)((s == 3 ? c2 : -c2) * (double)v2); }
            for (size_t i = 0; i < N; ++i) {
                upb[i] += hb[i];
                if (s) {
                    const real chi2 = (r->dW[i] + r->dZ[i] / sqrt3) / 2;
                    for (int j = 0; j < s; ++j) { kb[j][i] += dt * (real)T->A0[4 * s + j] * hb[i]; gb[j][i] += chi2 * (real)T->B0[4 * s + j] * hb[i]; }
                }
            }
            /* g_s = g(H1_s) */
            orc_f_backward(&h->cfg.diffusion, pg, h1, r->actg[s], B, 0, gb[s], hb, pbar + h->Pf);
            for (size_t i = 0; i < N; ++i) {
                upb[i] += hb[i];
                if (s) for (int j = 0; j < s; ++j) { kb[j][i] += dt * (real)T->A1[4 * s + j] * hb[i]; gb[j][i] += sqdt * (real)T->B1[4 * s + j] * hb[i]; }
            }
        }
        memcpy(U, upb, sizeof(real) * N);
    }
    if (nsave && h->save_t0)
        for (int c = 0; c < B; ++c)
            for (int i = 0; i < Dd; ++i) U[(size_t)c * Dd + i] += ubar[((size_t)c * nsave) * Dd + i];
    memcpy(xbar, U, sizeof(real) * N);
    free(U); free(upb); free(hb);
    for (int j = 0; j < 4; ++j) { free(kb[j]); free(gb[j]); }
    return 0;
}

/* rms(k4 - k3), rms(H0_4 - H0_3) of every ACCEPTED step of the last forward (eigen_est = n1 / n2), or of the last orc_sde_attempt (n_acc = 0 there). */
int orc_sde_eigen_norms(void* hh, real* n1n2) {
    sde_handle* h = (sde_handle*)hh;
    if (h->last_is_attempt || !h->have_tape) { n1n2[0] = h->last_nrm[0]; n1n2[1] = h->last_nrm[1]; return 0; }
    for (int n = 0; n < h->n_acc; ++n) { n1n2[2 * n] = h->rec[n].n1; n1n2[2 * n + 1] = h->rec[n].n2; }
    return h->n_acc;
}
