/*
 * rnde_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A from-scratch CPU restatement of the adaptive Tsit5 integration that sits
 * behind RegNeuralDE.jl's TrackedNeuralODE call operator
 * (reference: src/models/neural_ode.jl:48-180, `solve(prob, Tsit5(); ...)`).
 *
 * PARITY UNPINNED: the solver arithmetic of the reference lives in un-vendored
 * Julia packages (OrdinaryDiffEq 5.50.0, DiffEqBase 6.53.4 fork,
 * DiffEqCallbacks 2.16.0, Tracker 0.2.14 fork; reference Manifest.toml:242-278,
 * :964-968, :1326-1332).  None of them, and no Julia runtime, exist in the
 * build container, and the reference's own tests hold no golden vectors
 * (test/test_node.jl has no @test).  This file restates the *published*
 * algorithm (Tsitouras 2011 tableau; Hairer initial-step rule; PI controller)
 * as recalled in SURVEY.md Appendix A/B, and is pinned only by its own
 * self-tests (order conditions, convergence order, fp64 finite differences)
 * and by what third-party code in the build image can pin (tests/test_oracle.py):
 *   B.1 initial step     == scipy.integrate._ivp.common.select_initial_step on the
 *                           same f / tolerances / RMS norm (test_initdt_matches_hairer_scipy:
 *                           1e-12; the deviation list is that test's docstring -- it is empty
 *                           up to how the order argument is counted);
 *   B.2/B.3 step + EEst  == scipy rk_step / RK45.E / DOP853 E5 for the tableau-as-data
 *                           pairs (the Tsit5 table itself: order conditions only);
 *   B.4 PI controller    STRUCTURE == Hairer's published dopri5.f recurrence, and its
 *                           beta2 -> 0 limit == scipy's I controller
 *                           (test_pi_controller_is_hairers_dopri5_form); the EXPONENTS
 *                           beta1 = 7/(10 order), beta2 = 2/(5 order), gamma = 0.9,
 *                           qmin = 0.2, qmax = 10, qoldinit = 1e-4 are OrdinaryDiffEq
 *                           defaults as recalled: UNVERIFIABLE OFFLINE;
 *   B.5 callback at init, B.6 save_start, tracked-time end-of-interval handling:
 *                           UNVERIFIABLE OFFLINE (config flags below; tools/julia_golden.jl
 *                           dumps what a Julia host does).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (librnde.so) never links or calls it.
 *
 * Built twice from one source: real = float (librnde_oracle_f32.so) and
 * real = double (-DRNDE_F64, librnde_oracle_f64.so).
 */
#ifndef RNDE_ORACLE_H
#define RNDE_ORACLE_H

#ifdef RNDE_F64
typedef double real;
#else
typedef float real;
#endif

#define ORC_MAX_LAYERS 8

#ifdef __cplusplus
extern "C" {
#endif

/* Dynamics f(u,p,t): a Dense chain.
 *  time_dep=1: TDChain semantics (reference src/models/basic.jl:16-23 and
 *  experiments/mnist_node.jl:51-54): a row filled with t is vcat'ed onto the
 *  input of EVERY layer, so layer l has weight out_l x (in_l+1).
 *  pre_act=1: tanh applied to u before the first layer
 *  (experiments/latent_ode.jl:113-124).
 *  Parameter vector layout = Flux.destructure order: for each layer
 *  [vec(W) column-major (out x in_ext); b(out)]. */
typedef struct {
    int n_layers;
    int dims[ORC_MAX_LAYERS + 1]; /* dims[0] = D (state rows), dims[n_layers] = D */
    int act[ORC_MAX_LAYERS];      /* 0 identity, 1 tanh */
    int time_dep;
    int pre_act;
} orc_arch;

typedef struct {
    orc_arch arch;
    real reltol, abstol;
    int reg_kind;      /* 0 none (func ignored, neural_ode.jl:48-77); 1 EEst*dt (neural_ode.jl:116);
                          2 stiffness estimate (mnist_node.jl:74-79); 3 err + 0.1*stiff (mnist_node.jl:88-97) */
    int cb_save_start; /* 1: SavingCallback fires once at init with EEst=1, dt=0 -> pushes 0 [RECALL: unverifiable offline] */
    int track_ctrl;    /* 1: differentiate dt_next = dt/q through the controller on accepted steps [RECALL (Tracker overloads of DiffEqBase): unverifiable offline] */
    int track_initdt;  /* 1: differentiate the Hairer initial-step computation [values: == scipy's select_initial_step; that Tracker differentiates it: RECALL, unverifiable offline] */
    int max_attempts;
    int solver;        /* 0 Tsit5 (every reference call site), 1 DP5: Dormand-Prince 5(4), the second 7-stage FSAL pair of the
                          tableau-as-data path (validated against scipy's RK45) */
} orc_config;

int   orc_param_count(const orc_arch* a);
void* orc_create(const orc_config* cfg);
void  orc_destroy(void* h);

/* f evaluation, no tape: out[D x B] = f(u[D x B], p, t). column-major. */
void orc_f_eval(const orc_arch* a, const real* p, const real* u, int B, real t, real* out);

/* One Tsit5 attempt without controller (kernel-level parity):
 * given uprev, k1, t, dt -> k2..k7 (kout: 6 arrays D*B), unew, EEst. */
void orc_tsit5_attempt(const orc_config* cfg, const real* p, const real* uprev, const real* k1,
                       int B, real t, real dt, real* kout, real* unew, real* eest, real* eigen_est);

/* Hairer initial step. returns dt; f0 (D*B) optional out. */
real orc_initdt(const orc_config* cfg, const real* p, const real* u0, int B, real t0, real t1, real* f0_out);

/* Full solve, records a tape inside the handle.
 *  saveat/nsave: optional save times (return_multiple path, neural_ode.jl:79-108);
 *  u_out: D*B (nsave==0) or D*nsave*B laid out (D, T, B) column-major (utils.jl:17-19).
 *  saveval: caller buffer of max_attempts+1 reals; steps_log: 4 reals per attempt (t, dt, EEst, accepted).
 *  returns 0 ok, 1 max attempts exceeded, 2 dt underflow, 3 non-finite. */
int orc_forward(void* h, const real* x, const real* p, int B, real t0, real t1,
                const real* saveat, int nsave, real* u_out, long* nfe,
                real* saveval, int* nsaveval, real* steps_log, int* nattempts);

/* Replay: the NEXT orc_forward on this handle (and every one after it until orc_set_replay(h, 0, 0, 0)) takes attempt n with
 * the proposed step size dtp[n] (still clamped to t1 - t) and the accept decision acc[n], and stops after n attempts, instead
 * of following its own PI controller.  EEst, q11, q are still computed and recorded, and orc_backward differentiates the
 * recorded program as if the controller had produced the sequence.  Used to run the fp32 oracle, the fp64 oracle and the
 * device along ONE (t, dt) sequence at the reference tolerance, where each one's own sequence is set by rounding noise. */
void orc_set_replay(void* h, const real* dtp, const int* acc, int n);
/* 6 reals per attempt of the last forward: t, dt, dtp_in, EEst, accepted, q.  Returns the number of attempts. */
int  orc_steps_ext(void* h, real* out6, int cap);

/* Reverse pass of the recorded solve (discretise-then-optimise).
 *  ubar: cotangent of u_out (same shape); svbar: cotangent per saveval element.
 *  xbar[D*B], pbar[P], tspanbar[2]. */
int orc_backward(void* h, const real* ubar, const real* svbar, real* xbar, real* pbar, real* tspanbar);

/* Tableau access for unit tests. a: 7x7 row-major (a[s][j]), c[7], btilde[7]. */
void orc_tableau(double* a, double* c, double* btilde);
void orc_dense_weights(double theta, double* b7);
void orc_tableau_of(int solver, double* a, double* c, double* btilde);
void orc_dense_weights_of(int solver, double theta, double* b7);

#ifdef __cplusplus
}
#endif
#endif
