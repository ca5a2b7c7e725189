// rnde_node.h -- what the translation units of the C ABI over the ODE engines share: the handle, the allocation and error-check macros and the
// helpers one side offers the other (rnde.hip: creation, forward solves, debug / bench entry points; rnde_reverse.hip: the reverse passes, the
// classifier head and the optimiser steps).  Not an interface: include/rnde.h is.
#pragma once
#include "../../include/rnde.h"
#include "rnde_fwd.h"
#include "rnde_bwd.h"
#include "rnde_stage.h"
#include "rnde_bstage.h"
#include "rnde_stage_persist2.h"
#include "rnde_bstage_persist.h"
#include "rnde_solve_sync.h"
#include "rnde_binit_stage.h"
#include "rnde_head.h"
#include "rnde_chain.h"
#include "rk_tables.h"
#include "rnde_bchain.h"
#include "rnde_chainmw.h"
#include "rnde_bchainmw.h"

#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

// Every device allocation of the library goes through here.  RNDE_POISON=1 fills fresh memory with 0xFF bytes (NaN as float): a
// read of something that was never written then shows up as NaN in the results instead of depending on what the allocator hands
// back (a debugging aid; tests/test_gpu_edge.py runs a solve under it).
static hipError_t rnde_malloc(void** p, size_t bytes) {
    const hipError_t e = (hipMalloc)(p, bytes);
    const bool poison = getenv("RNDE_POISON") != nullptr;       // (read per allocation: allocations are rare)
    if (e == hipSuccess && poison && bytes) (void)hipMemset(*p, 0xFF, bytes);
    return e;
}
#define hipMalloc(p, n) rnde_malloc((void**)(p), (n))

// rnde_stage_solve.hip (its own translation unit, parameter blocks by address: the same struct definitions on both sides)
extern "C" hipError_t rnde_launch_stage_solve(const void* stage_params, const void* persist_sync, const void* solve_sync, int act2, int x3, hipStream_t s);
extern "C" hipError_t rnde_launch_x3_pack(const float* p, void* x3B, void* x3D, void* x3Bt, void* x3Dt, int D, int H, int MT, int WT, int R, int HT, hipStream_t s);
using namespace rnde;

struct rnde_node {
    rnde_node_config cfg{};
    int D = 0, H = 0, P = 0, BT = 8, act2 = 1;
    int Bpad_max = 0, nwg_max = 0;
    // stage engine (rnde_stage.h)
    int engine = 1;                       // 1 column-owner, 2 stage kernels, 3 chain engine (rnde_chain.h)
    ChainGeo cg{}; float* cfrags = nullptr; int NKD = 0, chain_alt = 0;
    size_t chain_lds_f = 0, chain_lds_b = 0;
    // multi-wave kernels of the chain engine (rnde_chainmw.h): 4 waves per 16 columns, activations taped in the slab by the forward
    rnde_comm* couple = nullptr; int couple_batch = 0, couple_world = 1;   // SURVEY 8e mode 2 (rnde_node_set_coupling)
    int rk_tab = 0; RkTab rk{};   // explicit RK pair as data: 1 = a 7-stage pair (DP5, or Tsit5 through the same path when RNDE_CHAIN_TAB=1), 2 = S stages (DOP853)
    // chain engine, multi-wave kernels: the whole adaptive solve as ONE launch (rnde_chainmw.h MW_SOLVE) while the tiles fit one XCD (<= 32)
    int mw_slot = 0;   // the XCD (blockIdx % 8) this handle's one-launch chain kernels work on while they fit one: handles take turns (process-wide counter)
    int mw_clean = 0, mw_retry_after = 8;   // non-sticky fallback of those kernels, as persist_clean / persist_retry_after of the stage engine
    int mw_solve = 1; unsigned long long* mw_xch = nullptr; unsigned* mw_xcc = nullptr; unsigned* mw_abort = nullptr; unsigned* h_mw_chk = nullptr; unsigned mw_epoch = 0;
    int mw_bsweep = 1; int* mw_bargs = nullptr; int* h_mw_bargs = nullptr; unsigned* h_mw_bchk = nullptr; bool pending_bsweep = false; int bsweep_nt = 0; bool bsweep_global = false;   // (the pending sweep's OWN tile count / meeting kind: a later forward may overwrite h->B before the verdict is read)
      // the reverse sweep as one launch (rnde_bchainmw.h SWEEP): per-attempt arguments [sv_lo | sv_hi | eig_c], check words
    int rk_S = 7, rk_order = 5;   // stages of the pair in first-same-as-last form (evaluations per attempted step = rk_S - 1), controller order
    int mw_lat = 0;               // the reference's latent-ODE shape (20 <-> 50, 8 layers): forward kernels with register-stationary weights
    int mw = 0; MwGeo mg{}; float* mw_tab = nullptr; float* mw_slab = nullptr; long long mw_slab_evals = 0; size_t mw_lds_f = 0, mw_lds_b = 0;
    float* cslab = nullptr; size_t cslab_floats = 0; float* ev_t = nullptr; float* h_ev_t = nullptr;   // chain reverse: (H, Z) dump, evaluation times
    int sMT = 0, sWT = 0, sR = 0, sHT = 0, sK2b = 0, sKHb = 0;
    f32x4 *spwB = nullptr, *spwD = nullptr, *spwBt = nullptr, *spwDt = nullptr;
    float* slab2 = nullptr;
    size_t stage_lds = 0;
    float* head_ws = nullptr; size_t head_ws_floats = 0;   // fused classifier head scratch
    // rnde_node_classifier_grad: work the forward enqueues BEHIND the copy its host wait needs (so it runs while the host wakes up),
    // the event that wait uses, the caller-independent buffers of the fused step, and "the reverse sweep's weight packs are already queued"
    std::function<rnde_status(hipStream_t)> after_solve; hipEvent_t ev_host = nullptr; bool rev_packed = false;
    float* cg_ws = nullptr; size_t cg_ws_floats = 0; std::vector<float> cg_sv;
    float* sv_t_dev = nullptr; size_t sv_cap = 0; std::vector<float> saveat;   // saveat times of the last forward
    float* replay_dev = nullptr; size_t replay_cap = 0; const float* replay_host = nullptr; int n_replay = 0;   // rnde_node_forward_replay (set for one forward)
    // persistent attempt kernel (rnde_stage_persist.h): 1 = in use, 0 = off (RNDE_PERSIST=0), -1 = disabled after a failure
    int wgrad_side_pct = 30, stage_generic = 0;
    int persist_clean = 0, persist_retry_after = 8, persist_fallbacks = 0;   // non-sticky fallback: clean multi-launch solves since the last failure, when to try again   // fixed at creation (config fields; RNDE_* environment overrides are read once, there)
    int persist2 = -1;   // two column tiles per workgroup in the forward attempt kernel: -1 automatic (by tile count), 0 never, 1 whenever possible (RNDE_PERSIST2, read at creation)
    int persist = 0, persist_spins = kPersistMaxSpins; int tslab_Bpad = -1; size_t tslab_bytes = 0; float* tslab = nullptr; unsigned *pabort = nullptr, *pxcc = nullptr; unsigned* h_pchk = nullptr;
    // the whole forward solve as one launch (rnde_stage_solve.h): 1 = use it where it applies, 0 = off (RNDE_STAGE_SOLVE=0 at creation); meeting granules, epoch of their tags
    int stage_solve = 1; unsigned long long* sxch = nullptr; unsigned s_epoch = 0; int one_launch_solves = 0;
    // the one-launch solve's Dense layers on the matrix cores (rnde_x3.h): 1 = on (RNDE_X3 at creation / rnde_node_set_matrix_mode), split weight images, "packed for the current p"
    int x3 = 0; void *x3B = nullptr, *x3D = nullptr, *x3Bt = nullptr, *x3Dt = nullptr; bool x3_packed = false, x3_fwd = false;      // x3_fwd: this forward's stage kernels run with x3 (weights split by its pack launch); x3_packed: the last forward ran the x3 solve, the four images hold ITS parameters (the reverse pass may use the transposed pair)
    hipStream_t wstream = nullptr;        // (experimental overlap path of the weight-gradient GEMMs)
    std::vector<hipEvent_t> wevents;
    // device
    float *f0 = nullptr, *h0 = nullptr, *u1 = nullptr, *f1 = nullptr, *h1 = nullptr, *arena = nullptr;
    float* xcopy = nullptr;  // private copy of x (the tape must not alias caller memory)
    long long arena_recs = 0, rec_stride = 0;
    float* pcopy = nullptr;
    StepState *ctl = nullptr, *ctl_final = nullptr;
    unsigned char *mbox = nullptr, *h_mbox = nullptr; size_t mbox_meta_off = 0;   // stage engine: everything the host reads per chunk, contiguous (one copy)
    StepMeta* meta = nullptr;
    InitRec* initrec = nullptr;
    float *errpart = nullptr, *initpart = nullptr;
    BwdBuffers bw{};
    // pinned host
    StepState* h_ctl = nullptr;
    StepMeta* h_meta = nullptr;
    InitRec* h_init = nullptr;
    float* h_scal = nullptr;
    // last forward
    int B = 0, Bpad = 0, nwg = 0, n_att = 0, predicted = 0;
    float t0 = 0, t1 = 0;
    bool have_tape = false;
    bool pending_bwd = false;
    float* diag_buf = nullptr;   // (RNDE_DIAG builds) cycle stamps   // an asynchronous reverse pass whose health words have not been looked at yet
    std::vector<int> sv_index;  // per attempt: index into saveval or -1
    int n_saveval = 0;
    // rnde_node_set_timing: HIP events around the attempt loop of the forward, the reverse sweep and the rest of the reverse pass
    int timing = 0; hipEvent_t tev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; bool tev_fwd = false, tev_bwd = false;
    std::string err;
};

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                           \
            return RNDE_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

static const rnde_status RNDE_INTERNAL_RETRY = static_cast<rnde_status>(100);
// ---- rnde.hip, used by rnde_reverse.hip ----
StepParams make_params(rnde_node* h, const float* x, int B, float t0, float t1, int tape);
MwParams make_mw_params(rnde_node* h, const StepParams& P);
ChainParams make_chain_params(rnde_node* h, const StepParams& P);
rnde_status couple_sum(rnde_node* h, float* partials, long long count, hipStream_t s);
hipError_t stage_pack(rnde_node* h, const float* p, f32x4* dst, int which, int MTrows, int Kb, hipStream_t st);
rnde_status stage_pack_all(rnde_node* h, const float* p_dev, hipStream_t s, const float* x_src = nullptr, long long x_floats = 0);
hipError_t slab_prepare(rnde_node* h, int Bpad, hipStream_t s);
void persist_check_enqueue(rnde_node* h, int grid, hipStream_t s);
bool persist_check_result(rnde_node* h, int C, int R, hipStream_t s);
bool bsweep_failed(rnde_node* h, hipStream_t s);
rnde_status forward_impl(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                         float* u_out_dev, const float* saveat_host, int32_t n_saveat, float* sv_out_dev,
                         int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream);
