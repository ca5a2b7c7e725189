/*
 * rnde_sde_oracle.h -- CPU ORACLE for the stochastic half of the path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A from-scratch CPU restatement of the adaptive stochastic Runge-Kutta integration behind RegNeuralDE.jl's
 * TrackedNeuralDSDE call operator (reference src/models/neural_sde.jl:44-146:
 * `solve(SDEProblem{false}(drift, diffusion, x, tspan, p), SOSRI(); sensealg = SensitivityADPassThrough(), callback, ...)`
 * with diagonal noise) and of the reverse sweep Tracker performs over it.
 *
 * PARITY UNPINNED.  The arithmetic lives in StochasticDiffEq 6.30.1 (src/perform_step/sri.jl, src/initdt.jl,
 * src/integrators/integrator_utils.jl, src/tableaus.jl, src/alg_utils.jl) and DiffEqNoiseProcess (RSwM3 stacks,
 * noise_interfaces/noise_process_interface.jl), none of which (nor a Julia runtime) exist in the build container; the
 * reference holds no golden vectors for this path.  What pins this file instead:
 *   - the SOSRI / SOSRI2 / SRIW1 tableaux satisfy Roessler's strong-order-1.5 conditions to 1e-12 (SOSRI2's beta4 row to
 *     1e-8: it is the output of a numerical optimisation) -- tests/test_sde_oracle.py;
 *   - strong convergence on SDEs with known solutions, exactness of the Brownian-bridge bookkeeping (every accepted path is
 *     a refinement of ONE Brownian path: the increments used sum to W(t1) reconstructed from the pool), fp64 finite
 *     differences of the reverse pass, nfe1 = nfe2 = 2 + 4 * attempts.
 * Each [RECALL] decision (controller constants, delta, the initial-step rule, "the controller strips tracking") is listed in
 * DESIGN.md section 3.2.  SDE sample paths depend on the random stream: the noise enters through an explicit POOL of
 * standard normals, so the oracle, the device and a Julia caller can be driven by the same draws.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef RNDE_SDE_ORACLE_H
#define RNDE_SDE_ORACLE_H
#include "rnde_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    orc_arch drift;       /* re1(p[1:len])(u): Dense chain, time independent (neural_sde.jl:45-47) */
    orc_arch diffusion;   /* re2(p[len+1:end])(u): Dense chain, output D (diagonal noise, neural_sde.jl:49-52) */
    real reltol, abstol;  /* experiments/mnist_nsde.jl:79-80: 1.4f-1 */
    int tableau;          /* 0 SOSRI (mnist_nsde.jl:49,:63), 1 SRIW1, 2 SOSRI2 */
    int reg_kind;         /* 0 none; 1 EEst*dt (neural_sde.jl:87; mnist_nsde.jl:48); 2 |eigen_est| / stability_size (mnist_nsde.jl:51-61: the shipped default) */
    int cb_save_start;    /* 1: the saving callback also fires at initialisation (EEst = 1, dt = 0 -> 0), as in the ODE oracle */
    int max_attempts;
    /* controller constants; 0 selects the recalled StochasticDiffEq defaults:
     * beta2 = 2/(5 order), beta1 = 7/(10 order) with order = 3/2; gamma = 9/10; qmin = 1/5; qmax = 1.125;
     * qoldinit = 1e-4; delta = 1 (1/6 for SRIW1) */
    real beta1, beta2, gamma, qmin, qmax, qoldinit, delta;
    real stability_size;  /* reg_kind 2: 0 selects StochasticDiffEq.alg_stability_size(SOSRI2()) = 10.6 [RECALL] */
} orc_sde_config;

/* tableau as data: lower-triangular 4x4 matrices row-major (row = stage), vectors of 4 */
typedef struct {
    double A0[16], A1[16], B0[16], B1[16], alpha[4], beta1[4], beta2[4], beta3[4], beta4[4], c0[4], c1[4];
    double order, delta;
} orc_sri_tableau;
int orc_sri_tableau_get(int id, orc_sri_tableau* out); /* 0 ok */

int   orc_sde_param_count(const orc_sde_config* cfg, int* len_drift);
void* orc_sde_create(const orc_sde_config* cfg);
void  orc_sde_destroy(void* h);

/* One attempt without controller (kernel-level parity): uprev, dt, dW, dZ (D*B each) -> k[4], g[4] (kg_out: 8 arrays D*B:
 * k1..k4, g1..g4), unew, EEst. */
void orc_sde_attempt(void* h, const real* p, const real* uprev, int B, real dt, const real* dW, const real* dZ,
                     real* kg_out, real* unew, real* eest);

/* Full solve.  x: D x B column-major; p = [p_drift; p_diffusion] (neural_sde.jl:17).
 * noise: pool of standard normals, n_pool draws of 2 * D * B reals each (xi_W block then xi_Z block, both D x B column-major).
 * Draw 0 makes the first increments (sqrt(dt) * xi); after that every accepted step that needs fresh or bridged noise and
 * every rejected step consumes the next draw, in order.  ndraws_out = draws consumed.
 * Returns 0 ok, 1 max attempts, 2 dt underflow, 3 non-finite, 4 noise pool exhausted.
 * steps_log: 4 reals per attempt (t, dt, EEst, accepted). */
int orc_sde_forward(void* h, const real* x, const real* p, int B, real t0, real t1, const real* noise, int n_pool,
                    real* u_out, long* nfe1, long* nfe2, real* saveval, int* nsaveval, real* steps_log, int* nattempts,
                    int* ndraws_out);
/* saveat: following forwards return u at every time of `saveat` (increasing, inside [t0, t1]) as a (D, T, B) column-major array
 * in u_out (the {R,true} call methods, neural_sde.jl:44-61,:84-113; diffeqsol_to_3dtrackedarray): linear interpolation inside a
 * step, u0 for a point equal to t0; orc_sde_backward then takes ubar of that shape.  n = 0: back to the end state. */
void orc_sde_set_saveat(void* h, const real* saveat, int n);
/* Replay (as orc_set_replay of the ODE oracle): following forwards take attempt n with step size dt[n] (still clamped to
 * t1 - t) and the accept decision acc[n], and stop after n attempts; the noise bookkeeping follows those decisions.
 * All-accepted equal steps = the fixed-step method (convergence tests); also freezes the sequence for finite differences. */
void orc_sde_set_replay(void* h, const real* dt, const int* acc, int n);
/* W(t1) - W(t0) and the Z analogue of the path the last forward walked (sum of the increments of its accepted steps). */
void orc_sde_path_total(void* h, real* w_total, real* z_total);

/* Reverse pass of the recorded solve: ubar (D x B), svbar per saveval element -> xbar (D x B), pbar (P).
 * Step sizes and noise increments are constants of the reverse pass (the SDE controller strips tracking: [RECALL]). */
int orc_sde_backward(void* h, const real* ubar, const real* svbar, real* xbar, real* pbar);

/* The two norms behind integrator.eigen_est, per ACCEPTED step of the last forward (2 reals each; returns the count), or -- after
 * orc_sde_attempt -- of that attempt (returns 0): n1 = rms(k4 - k3), n2 = rms(H0_4 - H0_3), eigen_est = n1 / n2. */
int orc_sde_eigen_norms(void* h, real* n1n2);

/* shared with rnde_oracle.c */
void orc_f_forward(const orc_arch* a, const real* p, const real* u, int B, real t, real* out, real* acts);
real orc_f_backward(const orc_arch* a, const real* p, const real* u, const real* acts, int B, real t, const real* kbar,
                    real* ubar_out, real* pbar);
int  orc_act_rows_total(const orc_arch* a);

#ifdef __cplusplus
}
#endif
#endif
