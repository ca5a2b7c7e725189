/*
 * rnde.h -- C ABI of librnde.so: the MI355X (gfx950) drop-in for the adaptive
 * Runge-Kutta integration behind RegNeuralDE.jl's TrackedNeuralODE call operator.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *   rnde_node_create    TrackedNeuralODE(model, tspan, time_dep, regularize, solver; kw...)
 *                       src/models/neural_ode.jl:10-33 (constructor; Flux.destructure at :12)
 *   rnde_node_forward   (n::TrackedNeuralODE{R,Z})(x, p; func, tspan, saveat)
 *                       src/models/neural_ode.jl:48-77 ({false,false}), :110-144 ({true,false}):
 *                       everything between ODEProblem construction (:128-129) and the unpacking
 *                       of sol.u[end] / sol.destats.nf / sv (:138-143), i.e. the `solve` call
 *                       (:131-137) that lives in OrdinaryDiffEq/DiffEqBase/DiffEqCallbacks.
 *   rnde_node_backward  the reverse sweep Tracker.gradient performs over that solve
 *                       (experiments/mnist_node.jl:229-232) because
 *                       sensealg = SensitivityADPassThrough() (neural_ode.jl:134).
 *   rnde_node_release_tape / rnde_node_destroy   Julia GC of the Tracker tape / the layer.
 *
 * All arrays are fp32, column-major, exactly as Julia holds them: x and u are D x B
 * (element (r,c) at c*D + r); p is the Flux.destructure vector
 * [vec(W1) (out x in_ext, column-major); b1; vec(W2); b2; ...] (neural_ode.jl:12).
 * Pointers named *_dev are DEVICE (HBM) pointers owned by the caller; the library keeps no
 * caller pointer after a call returns.  `stream` is a hipStream_t (0 = default stream); calls
 * are asynchronous on that stream except where they return host scalars (they synchronise
 * the stream before returning those).  One in-flight call per handle; handles are independent.
 * No exceptions cross this boundary: every function returns an rnde_status.
 */
#ifndef RNDE_H
#define RNDE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNDE_MAX_LAYERS 8

typedef enum {
    RNDE_OK = 0,
    RNDE_ERR_BAD_ARG = 1,       /* shape / config not supported                               */
    RNDE_ERR_MAX_ATTEMPTS = 2,  /* solver hit max_attempts (reference: maxiters, never checked) */
    RNDE_ERR_DT_UNDERFLOW = 3,  /* dt <= dtmin                                                 */
    RNDE_ERR_NONFINITE = 4,     /* NaN/Inf in EEst or dt                                       */
    RNDE_ERR_HIP = 5,           /* HIP runtime error, see rnde_last_error                      */
    RNDE_ERR_NO_TAPE = 6,       /* backward without a recorded forward                         */
    RNDE_ERR_NO_DEVICE = 7      /* no gfx950 device visible                                    */
} rnde_status;

typedef enum { RNDE_ACT_IDENTITY = 0, RNDE_ACT_TANH = 1 } rnde_act;
/* RNDE_SOLVER_TSIT5: every reference call site (experiments/mnist_node.jl:62-103, latent_ode.jl:131-136), all engines.
 * RNDE_SOLVER_DP5: Dormand-Prince 5(4), the second 7-stage first-same-as-last pair, through the tableau-as-data kernels of the chain
 * engine (Dense chains of width <= 64; callbacks none / EEst*dt): there a pair of that shape is a table, not a kernel. */
/* RNDE_SOLVER_DOP853: Hairer's 8(5,3) pair as a 13-stage first-same-as-last TABLE (csrc/rk_tables.h, generated from scipy's coefficients;
 * linear fifth-order error estimate, controller exponents for order 8): the same kernels with stage count, tape layout and evaluation
 * counts read from the table -- what a Verner pair (Vern7, 10 stages + the closing evaluation) would be once its
 * coefficients are at hand.  End state only (no dense output in the table); callbacks none / EEst*dt; NFE = 3 + 12 per attempted step. */
typedef enum { RNDE_SOLVER_TSIT5 = 0, RNDE_SOLVER_DP5 = 1, RNDE_SOLVER_DOP853 = 2 } rnde_solver;
/* func passed to the layer call (neural_ode.jl:116; experiments/mnist_node.jl:67,:74-79,:88-97) */
typedef enum { RNDE_REG_NONE = 0, RNDE_REG_ERR = 1, RNDE_REG_STIFF = 2, RNDE_REG_ERR_STIFF = 3,
               RNDE_REG_STIFF_DT = 4   /* |eigen_est * dt|: the callback of the reference's own test (test/test_node.jl:75,:84; the comment at src/models/neural_ode.jl:115) */
} rnde_reg;

typedef struct {
    /* dynamics: Dense chain; time_dep = TDChain semantics (src/models/basic.jl:16-23,
     * experiments/mnist_node.jl:51-54): a row of t is appended to every layer input */
    int32_t n_layers;
    int32_t dims[RNDE_MAX_LAYERS + 1]; /* dims[0] = dims[n_layers] = D */
    int32_t act[RNDE_MAX_LAYERS];
    int32_t time_dep;
    int32_t pre_act;       /* leading tanh (experiments/latent_ode.jl:114) */
    int32_t max_batch;     /* largest B any call will pass */
    int32_t solver;        /* rnde_solver */
    float   reltol, abstol;
    int32_t regularize;    /* rnde_reg: which value the saving callback records per accepted step */
    int32_t cb_save_start; /* 1: callback also fires at init (value 0 for RNDE_REG_ERR) -- SURVEY.md B.5 */
    int32_t track_ctrl;    /* 1: reverse pass differentiates dt_next = dt/q through the PI controller */
    int32_t track_initdt;  /* 1: reverse pass differentiates the initial-step heuristic */
    int32_t max_attempts;  /* tape capacity in attempted steps */
    int32_t device;        /* HIP device ordinal */
    int32_t col_tile;      /* 0 = auto; MNIST form: 16 (stage engine, the only engine of that form since round 4); small-width chains:
                            * 64 (chain engine: 16 columns per wave) */
    /* tuning (0 = default everywhere, so a zero-initialised tail keeps the defaults) */
    int32_t persist;        /* stage engine: 0 = one launch per attempted step where the shape allows, -1 = always the 7-launch kernels */
    int32_t wgrad_side_pct; /* share (per cent of the attempts) of the parameter-gradient GEMMs run beside the reverse sweep on the CUs
                             * it leaves idle: 0 = default (30), -1 = none */
    int32_t stage_generic;  /* 1 = never use the instantiations with the MNIST geometry (D = 784, H = 100) as compile-time constants */
} rnde_node_config;

typedef struct rnde_node rnde_node;
typedef struct rnde_comm rnde_comm;   /* gradient collective, see rnde_comm_* below */

const char* rnde_version(void);
const char* rnde_last_error(const rnde_node* h); /* h may be NULL: last create error */
int32_t     rnde_param_count(const rnde_node_config* cfg);

rnde_status rnde_node_create(const rnde_node_config* cfg, rnde_node** out);
void        rnde_node_destroy(rnde_node* h);

/* Forward solve on [t0,t1].  u_out_dev: D x B.  saveval_host (may be NULL when regularize==0):
 * room for max_attempts+1 floats.  keep_tape != 0 records what rnde_node_backward needs.
 * nfe_out = sol.destats.nf (neural_ode.jl:72,:142).  Synchronises `stream` before returning. */
rnde_status rnde_node_forward(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B,
                              float t0, float t1, float* u_out_dev, int64_t* nfe_out,
                              float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape,
                              void* stream);

/* The {R,true} call methods (reference src/models/neural_ode.jl:79-108, :146-180 and update_saveat! :35-46): same
 * solve, returning the state at every time of `saveat` (increasing, inside [t0,t1]) from the Tsit5 dense output.
 * u_saved_dev: D x n_saveat x B, column-major, exactly what diffeqsol_to_3dtrackedarray builds (src/utils.jl:17-19).
 * With a taped saveat forward, rnde_node_backward takes u_bar_dev of that same D x n_saveat x B shape. */
rnde_status rnde_node_forward_saveat(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B,
                                     float t0, float t1, const float* saveat_host, int32_t n_saveat,
                                     float* u_saved_dev, int64_t* nfe_out, float* saveval_host,
                                     int32_t* n_saveval_out, int32_t keep_tape, void* stream);

/* save_everystep = true (reference src/models/neural_ode.jl:10-11: the other way to `return_multiple`): the state after every accepted step, and
 * the initial state first when save_start != 0 -- sol_out_dev: D x n x B, column-major, n <= capacity returned in *n_out together with the times
 * (t_host_out: room for `capacity` floats, may be NULL).  The number of steps is known only after the solve, so the call solves twice (the step
 * sequence, then the same solve saving at those step ends: u_new itself, no interpolation); the rest as rnde_node_forward_saveat, including the
 * backward call afterwards (u_bar_dev: D x n x B).  More accepted steps than `capacity`: RNDE_ERR_BAD_ARG with *n_out set to the room needed. */
rnde_status rnde_node_forward_everystep(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                        int32_t save_start, float* sol_out_dev, int32_t capacity, float* t_host_out, int32_t* n_out,
                                        int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream);

/* Parity instrument: the same solve along a GIVEN sequence of attempts.  steps_host holds n_steps pairs
 * (dt_proposed, accepted != 0): attempt n runs with min(dt_proposed[n], t1 - t) and is accepted or rejected as told, and
 * the solve ends after n_steps attempts; the error estimate, q11 and q of every attempt are still computed and logged
 * (rnde_node_steps), the saving callback still fires per accepted step, the tape is recorded and rnde_node_backward
 * differentiates the recorded program as if the controller had produced the sequence.  At the reference tolerance
 * (reltol = abstol = 1.4e-8, experiments/mnist_node.jl:121-124) the fp32 error estimate sits on its rounding floor, so
 * two fp32 implementations choose different step sequences; replaying ONE sequence (e.g. the fp32 CPU solver's) is how
 * trajectories and gradients are compared element-wise there (tests/test_gpu_replay.py).  Not a reference entry point. */
rnde_status rnde_node_forward_replay(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B,
                                     float t0, float t1, const float* steps_host, int32_t n_steps,
                                     float* u_out_dev, int64_t* nfe_out, float* saveval_host,
                                     int32_t* n_saveval_out, int32_t keep_tape, void* stream);

/* Reverse pass of the last recorded forward.  u_bar_dev: D x B cotangent of u_out;
 * saveval_bar_host: one cotangent per saveval element (NULL = zeros).
 * Outputs: x_bar_dev (D x B), p_bar_dev (P, overwritten), tspan_bar_host[2] (may be NULL).
 * Synchronises `stream` before returning. */
rnde_status rnde_node_backward(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host,
                               float* x_bar_dev, float* p_bar_dev, float* tspan_bar_host,
                               void* stream);

/* The same reverse pass WITHOUT the final synchronisation: outputs are valid in stream order (x_bar_dev, p_bar_dev, and
 * tspan_bar_dev[2] when not NULL -- a DEVICE pointer here), the call returns as soon as the work is enqueued, so the caller
 * can queue the optimiser update and the next forward underneath the tail of this one (a training loop that never reads the
 * loss on the host has no reason to stop the GPU once per step).  A failure that only the GPU can report (a persistent
 * kernel abandoning a hand-off) surfaces as RNDE_ERR_HIP from the NEXT call on the handle. */
rnde_status rnde_node_backward_async(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host,
                                     float* x_bar_dev, float* p_bar_dev, float* tspan_bar_dev, void* stream);

rnde_status rnde_node_release_tape(rnde_node* h);

/* Host-pointer convenience variants (H2D/D2H copies inside; PCIe-inclusive). */
rnde_status rnde_node_forward_host(rnde_node* h, const float* x, const float* p, int32_t B, float t0,
                                   float t1, float* u_out, int64_t* nfe_out, float* saveval,
                                   int32_t* n_saveval_out, int32_t keep_tape);
rnde_status rnde_node_backward_host(rnde_node* h, const float* u_bar, const float* saveval_bar,
                                    float* x_bar, float* p_bar, float* tspan_bar);

/* Per-attempt log of the last forward: 4 floats per attempt (t, dt, EEst, accepted). */
rnde_status rnde_node_steps(rnde_node* h, float* steps_host, int32_t capacity, int32_t* n_attempts_out);

/* Kernel-level entry points used by the parity tests and by bench.py's roofline leg.
 * rnde_debug_attempt: ONE Tsit5 attempt (6 f evaluations + error estimate) from a given
 * (uprev, k1, t, dt); k_out_dev receives k2..k7 (6 x D x B), unew_out_dev D x B.
 * rnde_debug_feval: out = f(u, p, t).
 * rnde_bench_attempt: runs `iters` back-to-back launches of the step kernel on the handle's
 * workspace and returns the mean kernel time in microseconds measured with HIP events on
 * `stream` (the same kernel rocprofv3 reports as rnde_step_kernel). */
rnde_status rnde_debug_feval(rnde_node* h, const float* u_dev, const float* p_dev, int32_t B, float t,
                             float* out_dev, void* stream);
rnde_status rnde_debug_attempt(rnde_node* h, const float* uprev_dev, const float* k1_dev,
                               const float* p_dev, int32_t B, float t, float dt, float* k_out_dev,
                               float* unew_out_dev, float* eest_out, void* stream);
rnde_status rnde_bench_attempt(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B,
                               int32_t iters, float* mean_us_out, void* stream);
/* The same with the tape stores a training step's forward performs (keep_tape != 0): the kernel variant that dominates a
 * training step, which is what bench.py's roofline object is computed from. */
rnde_status rnde_bench_attempt_taped(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B,
                                     int32_t iters, float* mean_us_out, void* stream);
/* The same with a COLD tape: attempt i writes record i mod `records` of the arena (records >= 2), as the attempts of a solve do (31 records
 * of 21.7 MB at B = 512 per solve: the tape does not stay on the chip), where rnde_bench_attempt_taped rewrites record 0 for ever (it then
 * lives in the Infinity Cache).  Stage engine; this is the figure that agrees with the per-attempt time inside a training step. */
rnde_status rnde_bench_attempt_cold_tape(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters,
                                         int32_t records, float* mean_us_out, void* stream);
/* Measurement aid: with timing on, every forward / reverse records HIP events on the caller's stream around (a) the attempted
 * steps of the forward solve, (b) the reverse sweep, (c) the rest of the reverse pass (parameter-gradient GEMMs after the
 * sweep + reductions); rnde_node_timing waits for them and returns the three durations in ms (-1 = not recorded). */
rnde_status rnde_node_set_timing(rnde_node* h, int32_t on);
rnde_status rnde_node_timing(rnde_node* h, float* fwd_attempts_ms, float* rev_sweep_ms, float* rev_rest_ms);
int32_t     rnde_node_last_attempts(const rnde_node* h);   /* attempted steps of the last forward */
/* How often the one-launch attempt kernels gave up a hand-off (co-tenant on the CUs for > 1 s) and the handle went to the
 * 7-launch kernels; it returns to the one-launch kernels after 8, 16, 32, ... clean solves (not sticky). */
int32_t     rnde_node_fallback_count(const rnde_node* h);
/* How the handle currently runs one attempted step: number of kernel launches (1: rnde_stage_attempt_kernel /
 * rnde_step_kernel / rnde_chain_kernel; 7: rnde_stage_kernel START, 5 x STAGE, LAST) -- bench.py's roofline bookkeeping. */
int32_t     rnde_node_launches_per_attempt(const rnde_node* h);
/* How many forward solves of this handle ran as ONE kernel launch (attempt loop, PI controller and error-norm meeting inside the kernel:
 * rnde_stage_solve_kernel for the MNIST form at <= 512 columns, MW_SOLVE for the Dense-chain and SDE engines) -- the launch that replaces
 * the body of `solve(prob, Tsit5(); ...)`, reference src/models/neural_ode.jl:131-137.  bench.py's roofline bookkeeping and the tests. */
int32_t     rnde_node_one_launch_solves(const rnde_node* h);
/* Which arithmetic unit forms the Dense-layer products of the stage engine -- the f(u, p, t) evaluations inside `solve(prob, Tsit5(); ...)`, reference
 * src/models/neural_ode.jl:131-137 with the dynamics of experiments/mnist_node.jl:41-54, their transposes in the reverse pass and the parameter-gradient GEMMs:
 *   RNDE_MATRIX_F32   (0) the fp32-input MFMA v_mfma_f32_16x16x4_f32 -- on gfx950 an instruction of the VECTOR ALUs (64 FLOP/clk/SIMD, no overlap with
 *                         other vector work: tools/micro/coexec.hip); bit-identical to the launch-per-attempt kernels and mirrored by the oracle's
 *                         device-order mode;
 *   RNDE_MATRIX_BF16X3 (1, default where it applies: the MNIST form D = 784, H = 100 on the persistent kernels, any batch) both operands split EXACTLY into three bf16 numbers, the six
 *                         leading cross products on the matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulation: csrc/rnde_x3.h.  Closer to the fp64
 *                         restatement than mode 0 (fewer roundings per dot product), hence FEWER attempted steps at the reference tolerance, where
 *                         the step size is set by rounding noise; not bit-identical to mode 0.
 * The environment variable RNDE_X3 (0 / 1) sets the default of new handles.  Returns BAD_ARG for an unknown mode; a handle whose geometry the
 * bf16x3 kernels do not serve keeps mode 0 and says so through rnde_node_matrix_mode. */
#define RNDE_MATRIX_F32 0
#define RNDE_MATRIX_BF16X3 1
rnde_status rnde_node_set_matrix_mode(rnde_node* h, int32_t mode);
int32_t     rnde_node_matrix_mode(const rnde_node* h);

/* ======================================================================================================================
 * Several taped forwards alive at once behind one handle (the `tape_id` form of SURVEY.md 8b).  An rnde_node holds ONE tape; the
 * reference's loop sometimes needs more -- the NFE probe between a forward and its reverse (experiments/mnist_node.jl:245), two
 * batches in flight.  A tape pool is a set of solver instances of one configuration, created on demand (each taped one owns an
 * arena of max_attempts records): rnde_tapes_forward(keep_tape = 1) takes a free one and returns its index as the tape id,
 * rnde_tapes_backward(tape_id) or rnde_tapes_release(tape_id) frees it; keep_tape = 0 runs on an extra instance that never tapes
 * (tape id -1).  Arguments and results are those of rnde_node_forward / rnde_node_backward.
 * ====================================================================================================================== */
typedef struct rnde_tapes rnde_tapes;
rnde_status rnde_tapes_create(const rnde_node_config* cfg, int32_t max_tapes, rnde_tapes** out);
void        rnde_tapes_destroy(rnde_tapes* t);
const char* rnde_tapes_last_error(const rnde_tapes* t);     /* t may be NULL: last create error of this thread */
int32_t     rnde_tapes_in_use(const rnde_tapes* t);
rnde_node*  rnde_tapes_node(rnde_tapes* t, int32_t tape_id);  /* the instance behind a tape id (-1: the untaped one), for the rnde_node_* statistics / tuning calls */
rnde_status rnde_tapes_forward(rnde_tapes* t, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, float* u_out_dev,
                               int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream,
                               int32_t* tape_id_out);
rnde_status rnde_tapes_backward(rnde_tapes* t, int32_t tape_id, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                float* p_bar_dev, float* tspan_bar_host, void* stream);
rnde_status rnde_tapes_release(rnde_tapes* t, int32_t tape_id);

/* Fused caller of the hot path: postode Dense(D, C) + Flux.Losses.logitcrossentropy and their
 * reverse in one call -- replaces reference src/models/supervised_classification.jl:44-45 + experiments/mnist_node.jl:135
 * and the Tracker reverse of both.  p3 = Flux.destructure(Dense(D, C)) = [vec(W) (C x D col-major); b (C)];
 * y: one-hot C x B.  Outputs: logits (C x B, may be NULL), u_bar = d ce / d u (D x B), p3_bar (C*D + C),
 * ce (device scalar: mean cross entropy).  Asynchronous on `stream`. */
rnde_status rnde_classifier_head(rnde_node* h, const float* u_dev, const float* p3_dev, const float* y_dev,
                                 int32_t B, int32_t n_classes, float* logits_out_dev, float* u_bar_dev,
                                 float* p3_bar_dev, float* ce_out_dev, void* stream);

/* One training-step gradient of the reference's classifier loss in ONE call (experiments/mnist_node.jl:132-137 and :229-233:
 * `loss_function` + `Tracker.gradient` with ClassifierNODE, preode = identity):
 *     u = TrackedNeuralODE(x, p2)  ->  ce = logitcrossentropy(Dense_p3(u), y)  ->  loss = ce + lambda * mean(sv.saveval)
 * = rnde_node_forward (taped) + rnde_classifier_head + rnde_node_backward_async, with the head and the weight packs of the reverse
 * sweep queued BEFORE the forward's host wait, so the GPU does not idle while the host reads the step log and launches the sweep
 * (the three separate calls leave ~80 us of a 2.4 ms step idle at B = 512).  Stage engine only (two-layer dynamics, col_tile 0).
 * Outputs: p2_bar_dev (P), p3_bar_dev (C*D + C), x_bar_dev (D x B, may be NULL), ce_out_dev (device scalar), *reg_out_host =
 * lambda * mean(saveval) and *nfe_out (host, valid on return: the step log has been read); lambda = 0: no regulariser cotangent.
 * comm != NULL: both gradients are sum-all-reduced in place behind the reverse pass (rnde_comm_allreduce, mean = 0) -- in ONE call when
 * p3_bar_dev == p2_bar_dev + P (one flat buffer [p2-bar | p3-bar]), else in two.  Asynchronous from the reverse pass on, like
 * rnde_node_backward_async. */
rnde_status rnde_node_classifier_grad(rnde_node* h, const float* x_dev, const float* p2_dev, const float* p3_dev,
                                      const float* y_dev, int32_t B, int32_t n_classes, float t0, float t1, float lambda,
                                      float* p2_bar_dev, float* p3_bar_dev, float* x_bar_dev, float* ce_out_dev,
                                      float* reg_out_host, int64_t* nfe_out, rnde_comm* comm, void* stream);

/* Optimiser update of the reference's training step, one launch per parameter group:
 * Flux.Optimise.Optimiser(InvDecay(gamma), Momentum(eta, rho)) applied by update_parameters! -- replaces reference
 * experiments/mnist_node.jl:130 + src/utils.jl:149-156 for one flat group:
 *     g' = g / (1 + gamma * n);   v = rho * v - eta * g';   p = p + v          (n = the group's InvDecay counter, >= 1)
 * p, g, v: device vectors of `len` floats; in place on p and v.  No handle: pure function of its arguments, asynchronous. */
rnde_status rnde_momentum_step(float* p_dev, const float* g_dev, float* v_dev, int64_t len, int64_t n, float gamma,
                               float eta, float rho, void* stream);

/* The same update with the gradient scaled first: g' = gscale * g / (1 + gamma * n).  gscale = 1 / world folds the averaging of a
 * sum-all-reduced gradient (rnde_comm_allreduce with mean = 0) into the optimiser launch. */
rnde_status rnde_momentum_step_scaled(float* p_dev, const float* g_dev, float* v_dev, int64_t len, int64_t n, float gamma,
                                      float eta, float rho, float gscale, void* stream);
/* Flux.Optimise.ADAM(eta, (beta1, beta2)) (the optimiser of reference experiments/mnist_nsde.jl) on one flat parameter group, one launch:
 * m = beta1 m + (1 - beta1) g ; v = beta2 v + (1 - beta2) g^2 ; p -= eta (m / (1 - beta1^t)) / (sqrt(v / (1 - beta2^t)) + eps), with g
 * scaled by gscale first (1 / world after a sum-all-reduce).  t = 1, 2, ... is the step being taken; m, v start at zero. */
rnde_status rnde_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t len, int64_t t, float eta, float beta1,
                           float beta2, float eps, float gscale, void* stream);

/* ======================================================================================================================
 * Data parallelism: the one collective of a training step.  The reference is single-process; with the
 * minibatch sharded by columns over the GPUs of a node (one process per GPU, each integrating its shard with its own
 * controller), the only exchange is a sum of the flat gradient between Tracker.gradient and update_parameters!
 * (reference experiments/mnist_node.jl:229-233, src/utils.jl:149-156).  rnde_comm_* gives a non-Python caller that
 * collective from this library: RCCL over xGMI on the caller's stream.  RCCL is bound at run time (dlopen): no link-time
 * dependency, never loaded by single-GPU users.
 *   rank 0: rnde_comm_unique_id(id) -> ship the 128 bytes to the other ranks by any means (Julia: Distributed / MPI; Python:
 *   the torch.distributed store) -> every rank: rnde_comm_create(id, rank, world, device, &c) (collective: blocks until all
 *   ranks arrive) -> per step: rnde_comm_allreduce(c, grad_dev, n, mean, stream) -> rnde_comm_destroy(c).
 * ====================================================================================================================== */
#define RNDE_COMM_ID_BYTES 128
rnde_status rnde_comm_unique_id(uint8_t id_out[RNDE_COMM_ID_BYTES]);
rnde_status rnde_comm_create(const uint8_t id[RNDE_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, rnde_comm** out);
void        rnde_comm_destroy(rnde_comm* c);
int32_t     rnde_comm_world(const rnde_comm* c);
const char* rnde_comm_last_error(const rnde_comm* c);   /* c may be NULL: last create / id error of this thread */
/* The file the collective library was bound from (an RCCL already loaded in the process -- e.g. the copy PyTorch bundles -- is reused
 * rather than a second instance loaded beside it), or the load error. */
const char* rnde_comm_library(void);
/* In-place sum of n floats over the ranks (mean != 0: followed by a scale by 1 / world), asynchronous on `stream`.
 * MNIST-NODE payload: 166,418 floats = 665,672 B in ONE call (latency bound: one contiguous buffer). */
rnde_status rnde_comm_allreduce(rnde_comm* c, float* buf_dev, int64_t n, int32_t mean, void* stream);
/* `world` ranks that live in ONE process on ONE device (out[world]): same interface, the all-reduce is a one-workgroup kernel
 * per rank meeting the others through device memory (n <= 8192 floats).  What the coupled controller below is tested with on a
 * single GPU (two host threads, two handles), and what several shards per device would use. */
rnde_status rnde_comm_create_local_group(int32_t world, int32_t device, rnde_comm** out);
/* RNDE_OK unless an all-reduce of this communicator gave up waiting for a rank (in-process groups and the one-shot path; blocking:
 * call after the stream has been synchronised).  The coupled solves below check it themselves at their synchronisation points.
 * One-shot path: a time-out is STICKY -- the kernel that gave up and every later one fill their buffer with NaN instead of a partial sum,
 * and every rnde_comm_allreduce enqueued after the time-out became visible to the host returns RNDE_ERR_HIP (rnde_comm_last_error names the
 * limit in force: 20 s, RNDE_ONESHOT_TIMEOUT_MS overrides it).  Its all-reduces may be enqueued on different streams: a call on another
 * stream than the previous one is ordered behind it with an event (RCCL has no such constraint). */
rnde_status rnde_comm_health(rnde_comm* c);
/* One-shot all-reduce over peer-mapped windows (the 0.67 MB gradient message is latency bound, so each rank reads
 * the N - 1 peers' copies directly over its xGMI links in ONE kernel and sums them in rank order -- the same bits on every rank --
 * instead of walking a ring).  Every rank owns a window in its HBM, exported with hipIpcGetMemHandle and mapped by all peers
 * (world <= 16, one process per rank, all on one node).  OPT-IN until it has been measured on N > 1 GPUs; two ways in:
 *   RNDE_ONESHOT=1 in the environment of rnde_comm_create: the handles travel through the RCCL communicator that call builds (one
 *     64-byte all-gather); all-reduces of n <= 262,144 floats then take the one-shot kernel, larger ones ncclAllReduce;
 *   rnde_comm_window_create(device, &win, handle) on every rank -> ship the 64 bytes to all ranks by any means ->
 *     rnde_comm_create_peers(win, handles_in_rank_order, rank, world, &c) (takes ownership of win on success): no RCCL at all; larger buffers go
 *     in pieces of 262,144 floats.
 * rnde_comm_path says which path a communicator's all-reduces take. */
#define RNDE_COMM_WINDOW_BYTES 64
typedef struct rnde_comm_window rnde_comm_window;
rnde_status rnde_comm_window_create(int32_t device, rnde_comm_window** win_out, uint8_t handle_out[RNDE_COMM_WINDOW_BYTES]);
void        rnde_comm_window_destroy(rnde_comm_window* w);   /* only for a window that was NOT handed to rnde_comm_create_peers */
rnde_status rnde_comm_create_peers(rnde_comm_window* win, const uint8_t* handles, int32_t rank, int32_t world, rnde_comm** out);
const char* rnde_comm_path(const rnde_comm* c);

/* SURVEY.md 8e mode 2 -- ONE controller for all shards: with the minibatch split by columns over `world` handles, the error norm
 * that drives the step size is the RMS over ALL D x B_global entries (what a single-device run at B_global computes), so every
 * shard takes the same (dt, accept) sequence and the sharded run reproduces the single-device one up to the order of the sums.
 * After rnde_node_set_coupling(h, c, global_batch) every launch that produces per-workgroup partial sums of a batch-wide norm --
 * the two of the initial-step rule, each attempted step, each reversed attempt, the reverse of the initial-step rule -- is
 * followed by rnde_comm_allreduce(c, partials) on the solve's stream, the norms are means over D x global_batch, and the cotangents
 * of the saved values are multiplied by world (each rank passes the cotangent of ITS loss, whose data term is a mean over its own
 * columns: with this the usual average of the ranks' gradients is the gradient of the single-device loss).  One small collective
 * per attempted step: the parity option, not the fast path (default: independent controllers, rnde_node_set_coupling(h, NULL, 0)).
 * Every rank must make the same calls in the same order and hold the same number of columns (equal shards: the partial arrays are
 * summed element-wise).  Engines: the stage engine (MNIST-form networks) and the chain engine's multi-wave kernels (Dense chains of
 * width <= 64, one launch per attempt -- the one-launch solve keeps its controller in the kernel and is not used when coupled). */
rnde_status rnde_node_set_coupling(rnde_node* h, rnde_comm* c, int32_t global_batch);

/* ======================================================================================================================
 * TrackedNeuralDSDE: the stochastic layer (reference src/models/neural_sde.jl:1-146; caller ClassifierNSDE,
 * src/models/supervised_classification.jl:82-103; experiment experiments/mnist_nsde.jl:70-100).
 *   rnde_nsde_create    TrackedNeuralDSDE(model1, model2, tspan, regularize, solver; kw...)   neural_sde.jl:13-41
 *   rnde_nsde_forward / rnde_nsde_forward_saveat   (n::TrackedNeuralDSDE{R,false / true})(x, p; func): the `solve(SDEProblem{false}(...), SOSRI(); callback, ...)`
 *                       between :98 and :108 (and :54-56 for the unregularised method), returning what :109-113 unpack:
 *                       u (D x B), nfe1, nfe2 (the closures' counters, :46,:50) and the saved EEst*dt values
 *   rnde_nsde_backward  the reverse sweep Tracker performs over that solve (sensealg = SensitivityADPassThrough, :104)
 * p = vcat(destructure(model1), destructure(model2)) (:15-17): the drift chain's parameters, then the diffusion chain's.
 * Diagonal noise: the diffusion chain maps D -> D and multiplies the increments element-wise (:49-52).
 * Noise: a pool of standard normals, n_pool draws of 2*D*B floats each (xi_W block then xi_Z block, both D x B column-major),
 * consumed in order -- draw 0 makes the first increments, every later accepted step that needs fresh or bridged noise and
 * every rejected step takes the next one (at most 1 + attempts draws).  noise_dev = NULL: the library fills its own pool
 * from `seed` (Philox4x32-10 + Box-Muller); a caller that wants its own random stream (Julia: randn!) passes the pool.
 * ====================================================================================================================== */
typedef enum { RNDE_SDE_SOSRI = 0, RNDE_SDE_SRIW1 = 1, RNDE_SDE_SOSRI2 = 2 } rnde_sde_solver;

typedef struct {
    int32_t drift_layers;                       /* Dense chain of the drift, time independent (neural_sde.jl:45-47) */
    int32_t drift_dims[RNDE_MAX_LAYERS + 1];    /* drift_dims[0] = drift_dims[drift_layers] = D; every width <= 64 */
    int32_t drift_act[RNDE_MAX_LAYERS];         /* rnde_act per layer */
    int32_t diff_layers;                        /* Dense chain of the diffusion (neural_sde.jl:49-52) */
    int32_t diff_dims[RNDE_MAX_LAYERS + 1];
    int32_t diff_act[RNDE_MAX_LAYERS];
    int32_t max_batch;                          /* largest B (= batch x trajectories) any call will pass */
    int32_t solver;                             /* rnde_sde_solver: the tableau */
    float   reltol, abstol;                     /* experiments/mnist_nsde.jl:79-80 */
    int32_t regularize;                         /* what the saving callback records per accepted step: RNDE_REG_NONE, RNDE_REG_ERR (func = EEst*dt,
                                                 * neural_sde.jl:87, experiments/mnist_nsde.jl:48) or RNDE_REG_STIFF (|eigen_est| / stability_size,
                                                 * mnist_nsde.jl:51-61 -- what the shipped configs/mnist_nsde.yml:6 selects; SOSRI2 only:
                                                 * eigen_est = rms(k4 - k3) / rms(H0_4 - H0_3), the estimate AutoSOSRI2(SOSRI2()) fills) */
    int32_t cb_save_start;                      /* 1: the saving callback also fires at initialisation (pushes 0) */
    int32_t max_attempts;
    int32_t device;
    /* controller constants, 0 = the StochasticDiffEq defaults as recalled: beta2 = 2/(5 order),
     * beta1 = 7/(10 order), order = 3/2, gamma = 0.9, qmin = 0.2, qmax = 1.125, qoldinit = 1e-4, delta = 1 (SRIW1: 1/6) */
    float   beta1, beta2, gamma, qmin, qmax, qoldinit, delta;
    int32_t generic;                            /* 1 = never use the instantiations with the reference's shape (32 -> 64 -> 32, 32 -> 32)
                                                 * as compile-time constants (A/B runs, parity tests of the generic kernels) */
    float   stability_size;                     /* RNDE_REG_STIFF: the constant the estimate is divided by; 0 = StochasticDiffEq.alg_stability_size(SOSRI2())
                                                 * = 10.6 (mnist_nsde.jl:54-55) */
} rnde_nsde_config;

typedef struct rnde_nsde rnde_nsde;

int32_t     rnde_nsde_param_count(const rnde_nsde_config* cfg, int32_t* len_drift_out);
rnde_status rnde_nsde_create(const rnde_nsde_config* cfg, rnde_nsde** out);
void        rnde_nsde_destroy(rnde_nsde* h);
const char* rnde_nsde_last_error(const rnde_nsde* h);   /* h may be NULL: last create error of this thread */

/* Forward solve on [t0, t1].  x_dev, u_out_dev: D x B.  saveval_host: room for max_attempts + 1 floats (may be NULL when
 * regularize == 0).  nfe1_out = drift evaluations, nfe2_out = diffusion evaluations (2 + 4 per attempted step each).
 * keep_tape != 0 records what rnde_nsde_backward needs.  Synchronises `stream` before returning. */
rnde_status rnde_nsde_forward(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                              const float* noise_dev, int32_t n_pool, uint64_t seed, float* u_out_dev,
                              int64_t* nfe1_out, int64_t* nfe2_out, float* saveval_host, int32_t* n_saveval_out,
                              int32_t keep_tape, void* stream);
/* The {R,true} call methods (neural_sde.jl:44-61, :84-113; experiments/sde_toy_problem.jl:50-60): the same solve, returning the
 * state at every time of `saveat` (increasing, inside [t0, t1]) as the D x n_saveat x B array diffeqsol_to_3dtrackedarray builds
 * (src/utils.jl:17-19).  Points inside a step come from the SDE solution's linear interpolant (StochasticDiffEq has no
 * higher-order dense output), a save time equal to t0 is x.  rnde_nsde_backward then takes u_bar_dev of that D x n_saveat x B shape. */
rnde_status rnde_nsde_forward_saveat(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                     const float* noise_dev, int32_t n_pool, uint64_t seed, const float* saveat_host,
                                     int32_t n_saveat, float* u_saved_dev, int64_t* nfe1_out, int64_t* nfe2_out,
                                     float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream);

/* save_everystep = true of the SDE layer (reference src/models/neural_sde.jl:14): as rnde_node_forward_everystep -- the state after every accepted
 * step (t0 first when save_start != 0), D x n x B with n <= capacity in *n_out and the times in t_host_out (may be NULL); two solves on the same
 * noise inside.  rnde_nsde_backward afterwards takes the D x n x B cotangent. */
rnde_status rnde_nsde_forward_everystep(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                        const float* noise_dev, int32_t n_pool, uint64_t seed, int32_t save_start, float* sol_out_dev,
                                        int32_t capacity, float* t_host_out, int32_t* n_out, int64_t* nfe1_out, int64_t* nfe2_out,
                                        float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream);
/* Parity instrument (as rnde_node_forward_replay): the solve along n_steps given (dt, accepted != 0) pairs. */
rnde_status rnde_nsde_forward_replay(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                     const float* noise_dev, int32_t n_pool, const float* steps_host, int32_t n_steps,
                                     float* u_out_dev, int64_t* nfe1_out, int64_t* nfe2_out, float* saveval_host,
                                     int32_t* n_saveval_out, int32_t keep_tape, void* stream);
/* Reverse pass of the last recorded forward: u_bar_dev (D x B), saveval_bar_host (one per saveval element, NULL = zeros)
 * -> x_bar_dev (D x B), p_bar_dev (P, overwritten).  Step sizes and noise increments are constants of the reverse pass
 * (the SDE step-size controller strips tracking: DESIGN.md 3.2).  Synchronises `stream` before returning. */
rnde_status rnde_nsde_backward(rnde_nsde* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                               float* p_bar_dev, void* stream);
/* The same without the trailing synchronisation (nothing is read back on the host): the optimiser update and the next step can be
 * queued behind it on `stream`; u_bar / x_bar / p_bar must stay alive until the stream has passed them. */
rnde_status rnde_nsde_backward_async(rnde_nsde* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                               float* p_bar_dev, void* stream);
/* postsde Dense(D, C) + Flux.Losses.logitcrossentropy and their reverse for ClassifierNSDE with ONE trajectory per input (reference
 * src/models/supervised_classification.jl:96-97, experiments/mnist_nsde.jl): same contract as rnde_classifier_head, D = the SDE's state
 * dimension.  (With several trajectories the logits are averaged first: that stays on the host side.) */
rnde_status rnde_nsde_classifier_head(rnde_nsde* h, const float* u_dev, const float* p3_dev, const float* y_dev, int32_t B, int32_t n_classes,
                                      float* logits_out_dev, float* u_bar_dev, float* p3_bar_dev, float* ce_out_dev, void* stream);

/* The SDE counterpart of rnde_node_classifier_grad: one training-step gradient of ClassifierNSDE's loss around the solve, ONE trajectory
 * per input (reference src/models/supervised_classification.jl:82-103 with experiments/mnist_nsde.jl's loss_function and Tracker.gradient):
 *     u = TrackedNeuralDSDE(x, p2)  ->  ce = logitcrossentropy(Dense_p3(u), y)  ->  loss = ce + lambda * mean(sv.saveval)
 * = rnde_nsde_forward (taped; noise as there: the caller's pool, or NULL + seed for the library's stream) + rnde_nsde_classifier_head +
 * rnde_nsde_backward_async, the head queued before the forward's host wait.  x is the SDE's initial state (the presde layer's output), x_bar_dev
 * its cotangent (D x B, required: the presde layer's gradient needs it).  *reg_out_host, *nfe1_out, *nfe2_out are valid on return; the gradients and
 * ce_out_dev in stream order. */
rnde_status rnde_nsde_classifier_grad(rnde_nsde* h, const float* x_dev, const float* p2_dev, const float* p3_dev, const float* y_dev,
                                      int32_t B, int32_t n_classes, float t0, float t1, const float* noise_dev, int32_t n_pool,
                                      uint64_t seed, float lambda, float* p2_bar_dev, float* p3_bar_dev, float* x_bar_dev,
                                      float* ce_out_dev, float* reg_out_host, int64_t* nfe1_out, int64_t* nfe2_out, void* stream);
/* Per-attempt log of the last forward: 4 floats per attempt (t, dt, EEst, accepted); draws_out = noise draws consumed. */
rnde_status rnde_nsde_steps(rnde_nsde* h, float* steps_host, int32_t capacity, int32_t* n_attempts_out, int32_t* draws_out);
/* Kernel-level parity entry: ONE attempted step from (uprev, dt, dW, dZ), all D x B device arrays: kg_out_dev receives
 * k1..k4, g1..g4 (8 x D x B), unew_out_dev the proposed state, eest_out the error estimate. */
rnde_status rnde_nsde_debug_attempt(rnde_nsde* h, const float* uprev_dev, const float* p_dev, int32_t B, float dt,
                                    const float* dW_dev, const float* dZ_dev, float* kg_out_dev, float* unew_out_dev,
                                    float* eest_out, void* stream);
/* Measurement aid: HIP-event durations (ms, on the caller's stream) of the last forward's solve kernel (ONE launch = the
 * whole adaptive solve) and of the last reverse sweep kernel, with the attempt counts of that forward; -1 = not recorded. */
rnde_status rnde_nsde_timing(rnde_nsde* h, float* solve_ms, float* rev_sweep_ms, int32_t* attempts, int32_t* accepted);
/* Fill `n` floats with standard normals from (seed, stream_id): the library's own generator, exposed for its statistical test. */
rnde_status rnde_normal_fill(float* out_dev, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);

/* ======================================================================================================================
 * The latent-ODE caller of the hot path: recognition GRU, rec_to_gen + sampling, gen_to_data + masked likelihood
 * + KL, and their reverse passes -- what surrounds the layer call in LatentTimeSeriesModel (reference src/models/time_series.jl:40-70)
 * and loss_function (experiments/latent_ode.jl:206-236), at the reference's sizes: LatentGRU(37, 40, 50) (latent_ode.jl:39-106, :111),
 * rec_to_gen = Chain(Dense(100, 50, tanh), Dense(50, 40)) (:112), gen_to_data = Dense(20, 37) (:148).  One training step is
 *   rnde_latent_encode          x -> GRU over the time axis backwards (:99-106) -> rec_to_gen -> mu0, logvar, z0 = eps * exp(logvar / 2) + mu0
 *   rnde_node_forward_saveat    the layer call on z0 (the hot path, above)                                  (time_series.jl:61)
 *   rnde_latent_decode_loss     gen_to_data on every saved state, -mean(log_likelihood), mean(kl_divergence), and the reverse of the
 *                               likelihood term: res-bar (the u_bar of rnde_node_backward) and p4-bar          (latent_ode.jl:192-204, :226-233)
 *   rnde_node_backward          -> z0-bar
 *   rnde_latent_encode_backward z0-bar and lambda_k * mean(KL) -> p1-bar (GRU), p2-bar (rec_to_gen)
 * Layouts are Julia's: x is (2 * 37 + 1) x T x B = vcat(data, mask, dt) (latent_ode.jl:225), res is 20 x T x B, the parameter vectors are
 * Flux.destructure's (time_series.jl:11-14).  eps is the caller's standard-normal sample, 20 x B (CUDA.randn, time_series.jl:58).
 * All calls are asynchronous on `stream`; the handle keeps the tapes of ONE encode at a time.
 * ====================================================================================================================== */
typedef struct { int32_t max_batch, max_T, device; } rnde_latent_config;   /* max_T <= 64 save times */
typedef struct rnde_latent rnde_latent;
rnde_status rnde_latent_create(const rnde_latent_config* cfg, rnde_latent** out);
void        rnde_latent_destroy(rnde_latent* h);
const char* rnde_latent_last_error(const rnde_latent* h);      /* h may be NULL: last create error of this thread */
void        rnde_latent_param_counts(int32_t* n_p1, int32_t* n_p2, int32_t* n_p4);      /* 29,320 / 7,090 / 777 */
rnde_status rnde_latent_encode(rnde_latent* h, const float* x_dev, const float* p1_dev, const float* p2_dev, const float* eps_dev, int32_t B,
                               int32_t T, float* z0_out_dev, float* mu0_out_dev, float* logvar_out_dev, void* stream);
/* loss2_out_dev[0] = -mean_b(ll_b), [1] = mean_b(KL_b); res_bar_out_dev (20 x T x B) and p4_bar_out_dev (777) are the cotangents of term [0]. */
rnde_status rnde_latent_decode_loss(rnde_latent* h, const float* res_dev, const float* p4_dev, const float* x_dev, int32_t B, int32_t T,
                                    float* loss2_out_dev, float* res_bar_out_dev, float* p4_bar_out_dev, void* stream);
/* z0_bar_dev: 20 x B from the layer's reverse pass; the KL term enters with weight lambda_k (latent_ode.jl:178, :232). */
/* Flux.Optimise.Optimiser(InvDecay(gamma), AdaMax(eta, (beta1, beta2))) on one flat parameter group, in place (experiments/latent_ode.jl:108).
 * m, u: the caller's state arrays (zeros at the first step); n: the InvDecay counter (1 at the first step); beta1_pow: Flux's running
 * beta1^t (beta1 at the first step) -- the caller advances n and beta1_pow.  Asynchronous on `stream`. */
rnde_status rnde_adamax_step(float* p_dev, const float* g_dev, float* m_dev, float* u_dev, int64_t len, int64_t n, float gamma, float eta, float beta1,
                             float beta2, float eps, float beta1_pow, void* stream);
rnde_status rnde_latent_encode_backward(rnde_latent* h, const float* z0_bar_dev, float lambda_k, const float* p1_dev, const float* p2_dev,
                                        const float* x_dev, float* p1_bar_out_dev, float* p2_bar_out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
