// rnde.hip -- C ABI (include/rnde.h) over the gfx950 kernels.  No torch, no oracle, no CPU fallback:
// every entry point either runs the HIP kernels or returns an error status.
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
extern "C" hipError_t rnde_launch_stage_solve(const void* stage_params, const void* persist_sync, const void* solve_sync, int act2, hipStream_t s);
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
    int mw_bsweep = 1; int* mw_bargs = nullptr; int* h_mw_bargs = nullptr; unsigned* h_mw_bchk = nullptr; bool pending_bsweep = false;   // the reverse sweep as one launch (rnde_bchainmw.h SWEEP): per-attempt arguments [sv_lo | sv_hi | eig_c], check words
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

static thread_local std::string g_create_err;   // last create error of the calling thread (rnde_last_error(NULL)); no process-wide mutable state
static rnde_status chain_bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                 float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync = true, float* tspan_bar_dev = nullptr);

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                           \
            return RNDE_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

extern "C" const char* rnde_version(void) { return "rnde 0.1.0 (gfx950)"; }
extern "C" const char* rnde_last_error(const rnde_node* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int32_t rnde_param_count(const rnde_node_config* c) {
    int n = 0;
    for (int l = 0; l < c->n_layers; ++l) n += (c->dims[l] + (c->time_dep ? 1 : 0)) * c->dims[l + 1] + c->dims[l + 1];
    return n;
}

static StepParams make_params(rnde_node* h, const float* x, int B, float t0, float t1, int tape) {
    StepParams P{};
    P.x = x;
    P.f0 = h->f0; P.h0 = h->h0; P.u1 = h->u1; P.f1 = h->f1; P.h1 = h->h1;
    P.arena = h->arena; P.rec_stride = h->rec_stride;
    P.ctl = h->ctl; P.ctl_final = h->ctl_final; P.meta = h->meta; P.initrec = h->initrec;
    P.errpart = h->errpart; P.initpart = h->initpart; P.dbg_out = nullptr;
    P.D = h->D; P.H = h->H; P.B = B; P.Bn = h->couple ? h->couple_batch : B;
    P.Bpad = ((B + 15) / 16) * 16;   // both engines pad the batch to 16 columns (one tape format)
    P.nwg = h->engine == 2 ? h->sR * (P.Bpad / 16) : (h->engine == 3 ? (h->mw ? P.Bpad / 16 : (P.Bpad / 16 + kCW - 1) / kCW) : P.Bpad / h->BT);
    P.reltol = h->cfg.reltol; P.abstol = h->cfg.abstol; P.t0 = t0; P.t1 = t1;
    P.tape = tape; P.max_attempts = h->cfg.max_attempts;
    P.forced = 0; P.forced_t = 0; P.forced_dt = 0;
    P.xvec = ((h->D & 3) == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0;
    P.reg_kind = h->cfg.regularize;
    P.replay = nullptr; P.n_replay = 0;
    P.beta1 = (float)(7.0 / (10.0 * h->rk_order)); P.beta2 = (float)(2.0 / (5.0 * h->rk_order)); P.rk_order = (float)h->rk_order;   // (order 5: kBeta1, kBeta2 bit for bit)
    return P;
}

static_assert(kCMaxL == RNDE_MAX_LAYERS, "chain engine layer limit");
static bool chain_geo(const rnde_node_config* c, ChainGeo& G) {
    G = ChainGeo{};
    G.n_layers = c->n_layers; G.time_dep = c->time_dep ? 1 : 0; G.pre_act = c->pre_act ? 1 : 0;
    int po = 0, fo = 0, bo = 0, to = 0;
    for (int l = 0; l <= c->n_layers; ++l) {
        if (c->dims[l] < 1 || c->dims[l] > 4 * kCMaxKs) return false;
        G.width[l] = c->dims[l]; G.nks[l] = (c->dims[l] + 3) / 4;
    }
    for (int l = 0; l < c->n_layers; ++l) {
        G.act[l] = c->act[l];
        G.poff[l] = po; po += (G.width[l] + G.time_dep) * G.width[l + 1] + G.width[l + 1];
        G.foff[l] = fo; fo += ((G.nks[l + 1] + 3) / 4) * G.nks[l];
        G.boff[l] = bo; bo += 4 * ((G.nks[l + 1] + 3) / 4) * (1 + G.time_dep);   // padded to whole tiles
        G.toff[l] = to; to += ((G.nks[l] + 3) / 4) * G.nks[l + 1];
    }
    G.nfrag_f = fo; G.nfrag_b = bo; G.nfrag_t = to; G.nksD = G.nks[0];
    return true;
}
static rnde_status chain_create(const rnde_node_config* c, rnde_node** out) {
    ChainGeo G;
    if (!chain_geo(c, G)) { g_create_err = "unsupported dynamics: beyond the 2-layer time-dependent form the kernels cover Dense chains of width <= 64"; return RNDE_ERR_BAD_ARG; }
    // LDS: fragment tables rounded up to whole 1 KiB DMA units, then 64 floats of reduction scratch
    size_t lds_f = ((size_t)((G.nfrag_f + G.nfrag_b + 3) / 4) * 256 + 64) * 4, lds_b = ((size_t)((G.nfrag_f + G.nfrag_b + G.nfrag_t + 3) / 4) * 256 + 64) * 4;
    if (lds_b > 160 * 1024) { g_create_err = "chain too large: its weight fragments must fit the 160 KB LDS of a CU"; return RNDE_ERR_BAD_ARG; }
    if (c->regularize < RNDE_REG_NONE || c->regularize > RNDE_REG_ERR_STIFF) { g_create_err = "regularize: unknown value"; return RNDE_ERR_BAD_ARG; }
    if (c->col_tile != 0 && c->col_tile != 64 && c->col_tile != 65) { g_create_err = "col_tile: this network runs on the chain engine (0 = auto, 64 = one wave per column tile, 65 = four waves per column tile)"; return RNDE_ERR_BAD_ARG; }
    if (c->max_batch < 1 || c->max_attempts < 1) { g_create_err = "max_batch / max_attempts"; return RNDE_ERR_BAD_ARG; }
    rnde_node* h = new rnde_node();
    h->cfg = *c; h->engine = 3; h->cg = G;
    h->D = c->dims[0]; h->H = 0; h->P = rnde_param_count(c); h->BT = 16;
    h->NKD = G.nksD <= 4 ? 4 : (G.nksD <= 8 ? 8 : 16);
    {   // compile-time shape specialisation for the reference's own latent-ODE widths (rnde_chain.h: ALT)
        bool alt = G.nksD == kAltA && !getenv("RNDE_CHAIN_GENERIC");
        for (int l = 0; l <= G.n_layers && alt; ++l) alt = G.nks[l] == ((l & 1) ? kAltB : kAltA);
        h->chain_alt = alt ? 1 : 0;
    }
    h->Bpad_max = ((c->max_batch + 15) / 16) * 16;
    const int ntiles = h->Bpad_max / 16;
    h->nwg_max = (ntiles + kCW - 1) / kCW;
    {   // multi-wave kernels: geometry, LDS budget, who runs (col_tile 0 = auto -> multi-wave, 64 = one wave per tile, 65 = multi-wave)
        MwGeo& M = h->mg;
        M = MwGeo{};
        M.n_layers = G.n_layers; M.time_dep = G.time_dep; M.pre_act = G.pre_act; M.D = c->dims[0];
        int fo = 0, to = 0, row = 0;
        for (int l = 0; l <= G.n_layers; ++l) { M.width[l] = G.width[l]; M.mt[l] = (G.width[l] + 15) / 16; }
        for (int l = 0; l < G.n_layers; ++l) {
            M.act[l] = G.act[l]; M.poff[l] = G.poff[l];
            M.foff[l] = fo; fo += M.mt[l] * M.mt[l + 1] * 4;
            M.toff[l] = to; to += M.mt[l] * M.mt[l + 1] * 4;
            M.hrow[l] = row; row += 4 * M.mt[l]; M.zrow[l] = row; row += 4 * M.mt[l + 1];
        }
        M.hrow[G.n_layers] = row; row += 4 * M.mt[G.n_layers];
        M.RS = row; M.nfrag_f = (fo + 3) / 4 * 4; M.nfrag_t = (to + 3) / 4 * 4;
        h->mw_lds_f = ((size_t)M.nfrag_f * 64 + 1024 + 2048 + 64) * 4;
        h->mw_lds_b = ((size_t)M.nfrag_t * 64 + 1024 + 2048 + 64) * 4;
        const bool fits = h->mw_lds_f <= 160 * 1024 && h->mw_lds_b <= 160 * 1024;
        const char* e = getenv("RNDE_CHAIN_MW");
        h->mw = (fits && c->col_tile != 64 && !(e && e[0] == '0')) ? 1 : 0;
        if (c->col_tile == 65 && !fits) { g_create_err = "col_tile 65: the multi-wave kernels need the padded weight fragments in 160 KB of LDS"; delete h; return RNDE_ERR_BAD_ARG; }
        if (h->mw) h->nwg_max = ntiles;
        {   // experiments/latent_ode.jl:113-124 exactly: eight time-independent layers of widths 20 <-> 50
            bool lat = h->mw && c->n_layers == kLatLayers && !c->time_dep && h->NKD == 8;
            for (int i = 0; i <= kLatLayers && lat; ++i) lat = c->dims[i] == lat_width(i);      // (the kernels leave out the k-steps of the padding: exact widths)
            const char* e = getenv("RNDE_CHAIN_LAT");
            h->mw_lat = (lat && !(e && e[0] == '0')) ? 1 : 0;
        }
        if ((c->solver == RNDE_SOLVER_DP5 || c->solver == RNDE_SOLVER_DOP853) && !h->mw) { g_create_err = "DP5 / DOP853 need the multi-wave kernels (weights must fit LDS)"; delete h; return RNDE_ERR_BAD_ARG; }
        h->rk_tab = (h->mw && (c->solver == RNDE_SOLVER_DP5 || getenv("RNDE_CHAIN_TAB") != nullptr)) ? 1 : 0;
        if (c->solver == RNDE_SOLVER_DOP853) {
            // an S-stage pair as a table (csrc/rk_tables.h): the kernels take stage count, tape layout and evaluation counts from it.  Neither a
            // dense output nor the stiffness estimate (k_S - k_{S-1} is not an eigenvalue estimate for an arbitrary pair) exist for it.
            if (c->regularize >= 2) { g_create_err = "DOP853: callbacks none / error_est only (no stiffness estimate for a table pair)"; delete h; return RNDE_ERR_BAD_ARG; }
            h->rk_tab = 2; h->mw_lat = 0; h->rk_S = kDop853S; h->rk_order = kDop853Order;
            RkTab& T = h->rk;
            T = RkTab{};
            T.S = h->rk_S; T.order = h->rk_order;
            for (int sI = 0; sI < T.S; ++sI) {
                T.c[sI] = (float)kDop853C[sI]; T.bt[sI] = (float)kDop853Bt[sI];
                for (int i = 0; i < kRkSMax - 1; ++i) {
                    T.fwd[sI][i] = (sI + 1 + i < T.S) ? (float)kDop853A[sI + 1 + i][sI] : 0.f;
                    T.bwd[sI][i] = (sI - 1 - i >= 0) ? (float)kDop853A[sI][sI - 1 - i] : 0.f;
                }
            }
            for (int j = 0; j < T.S; ++j) T.aN[j] = (float)kDop853A[T.S - 1][j];
        }
        if (h->rk_tab == 1) {
            double A[7][7] = {{0}}, Cn[7], BT[7], Dn[7][4];
            if (c->solver == RNDE_SOLVER_DP5) {   // Dormand & Prince 1980; dense output: Shampine 1986 (the matrix scipy's RK45 uses)
                const double a[7][7] = {{0}, {1.0 / 5}, {3.0 / 40, 9.0 / 40}, {44.0 / 45, -56.0 / 15, 32.0 / 9}, {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
                                        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}, {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0}};
                const double cc[7] = {0, 0.2, 0.3, 0.8, 8.0 / 9, 1, 1};
                const double bt[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
                const double dn[7][4] = {{1.0, -8048581381.0 / 2820520608.0, 8663915743.0 / 2820520608.0, -12715105075.0 / 11282082432.0}, {0, 0, 0, 0},
                                         {0, 131558114200.0 / 32700410799.0, -68118460800.0 / 10900136933.0, 87487479700.0 / 32700410799.0},
                                         {0, -1754552775.0 / 470086768.0, 14199869525.0 / 1410260304.0, -10690763975.0 / 1880347072.0},
                                         {0, 127303824393.0 / 49829197408.0, -318862633887.0 / 49829197408.0, 701980252875.0 / 199316789632.0},
                                         {0, -282668133.0 / 205662961.0, 2019193451.0 / 616988883.0, -1453857185.0 / 822651844.0},
                                         {0, 40617522.0 / 29380423.0, -110615467.0 / 29380423.0, 69997945.0 / 29380423.0}};
                memcpy(A, a, sizeof(A)); memcpy(Cn, cc, sizeof(Cn)); memcpy(BT, bt, sizeof(BT)); memcpy(Dn, dn, sizeof(Dn));
            } else {   // Tsit5 through the data path (cross-check of the path itself): tableau of rnde_device.h, dense output expanded to monomials
                for (int sI = 0; sI < 7; ++sI) { Cn[sI] = tsC(sI); BT[sI] = tsBt(sI); for (int j = 0; j < 7; ++j) A[sI][j] = tsA(sI, j); }
                auto mul = [](const double* a, int na, const double* b, int nb, double* o) { for (int i = 0; i < na + nb - 1; ++i) o[i] = 0; for (int i = 0; i < na; ++i) for (int j = 0; j < nb; ++j) o[i + j] += a[i] * b[j]; };
                auto put = [&](int i, double cf, const double* poly5) { for (int j = 0; j < 4; ++j) Dn[i][j] = cf * poly5[j + 1]; };   // poly5[k] = coefficient of theta^k, k = 0..4
                {   // b1 = c th (th - r)(th^2 - p th + q)
                    const double f1[2] = {-1.3299890189751412, 1.0}, f2[3] = {0.7139816917074209, -1.4364028541716351, 1.0}; double t3[4], t5[5], th[2] = {0.0, 1.0};
                    mul(f1, 2, f2, 3, t3); mul(th, 2, t3, 4, t5); put(0, -1.0530884977290216, t5);
                }
                const double cq[2] = {0.1017, 2.490627285651252793}, pq[2] = {2.1966568338249754, 2.38535645472061657}, qq[2] = {1.2949852507374631, 1.57803468208092486};
                for (int i = 0; i < 2; ++i) { const double t5[5] = {0, 0, qq[i], -pq[i], 1.0}; put(1 + i, cq[i], t5); }   // c th^2 (th^2 - p th + q)
                const double c4[4] = {-16.54810288924490272, 47.37952196281928122, -34.87065786149660974, 2.5}, r4[4] = {1.21712927295533244, 1.203071208372362603, 1.2, 1.0},
                             s4[4] = {0.61620406037800089, 0.658047292653547382, 0.666666666666666667, 0.6};
                for (int i = 0; i < 4; ++i) { const double t5[5] = {0, 0, r4[i] * s4[i], -(r4[i] + s4[i]), 1.0}; put(3 + i, c4[i], t5); }   // c (th - r)(th - s) th^2
            }
            RkTab& T = h->rk;
            T = RkTab{};
            T.S = 7; T.order = 5;
            for (int sI = 0; sI < 7; ++sI) {
                T.c[sI] = (float)Cn[sI]; T.bt[sI] = (float)BT[sI];
                for (int i = 0; i < 6; ++i) { T.fwd[sI][i] = (sI + 1 + i < 7) ? (float)A[sI + 1 + i][sI] : 0.f; T.bwd[sI][i] = (sI - 1 - i >= 0) ? (float)A[sI][sI - 1 - i] : 0.f; }
                for (int j = 0; j < 4; ++j) T.dense[sI][j] = (float)Dn[sI][j];
            }
            for (int j = 0; j < 7; ++j) T.aN[j] = (float)A[6][j];
        }
    }
    h->chain_lds_f = lds_f; h->chain_lds_b = lds_b;
    if (hipSetDevice(c->device) != hipSuccess) { g_create_err = "hipSetDevice failed"; delete h; return RNDE_ERR_HIP; }
    const size_t Ac = (size_t)ntiles * h->NKD * 64;   // fragment-order arrays are padded to NKD k-steps
    h->rec_stride = ChainRec{(long long)Ac, h->rk_S}.total();
    auto dm = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    bool ok = true;
    ok &= dm((void**)&h->f0, Ac * 4) && dm((void**)&h->u1, Ac * 4) && dm((void**)&h->f1, Ac * 4) && dm((void**)&h->xcopy, (size_t)h->D * h->Bpad_max * 4);
    ok &= dm((void**)&h->pcopy, (size_t)h->P * 4) && dm((void**)&h->cfrags, (size_t)(G.nfrag_f + G.nfrag_b + G.nfrag_t + 4) * 256);
    if (h->mw) ok &= dm((void**)&h->mw_tab, mw_tab_floats(h->mg) * 4);
    if (h->mw) {
        const size_t xb = (size_t)(c->max_attempts + 4) * 3 * kMwMeetMax * 8;
        ok &= dm((void**)&h->mw_xch, xb) && dm((void**)&h->mw_xcc, kMwMeetMax * 4) && dm((void**)&h->mw_abort, 8);
        ok &= hipHostMalloc((void**)&h->h_mw_chk, (kMwMeetMax + 8) * 4) == hipSuccess;
        ok &= hipHostMalloc((void**)&h->h_mw_bchk, (kMwMeetMax + 8) * 4) == hipSuccess && dm((void**)&h->mw_bargs, (size_t)c->max_attempts * 16) &&
              hipHostMalloc((void**)&h->h_mw_bargs, (size_t)c->max_attempts * 16) == hipSuccess;
        if (ok) memset(h->h_mw_bchk, 0, (kMwMeetMax + 8) * 4);
        { static std::atomic<int> next_slot{0}; h->mw_slot = next_slot.fetch_add(1) & 7; }
        if (const char* eb = getenv("RNDE_CHAIN_BSWEEP")) h->mw_bsweep = atoi(eb);
        if (ok) { hipMemset(h->mw_xch, 0, xb); hipMemset(h->mw_abort, 0, 8); }
        const char* e = getenv("RNDE_CHAIN_SOLVE");
        if (e && e[0] == '0') h->mw_solve = 0;
    }
    ok &= dm((void**)&h->ctl, 2 * sizeof(StepState)) && dm((void**)&h->ctl_final, sizeof(StepState));
    ok &= dm((void**)&h->meta, (size_t)(c->max_attempts + 1) * sizeof(StepMeta)) && dm((void**)&h->initrec, sizeof(InitRec));
    ok &= dm((void**)&h->errpart, (size_t)(6 * h->nwg_max + 256) * 4) && dm((void**)&h->initpart, (size_t)(3 * h->nwg_max + 256) * 4);   // (+256: sum_partials reads whole 256-entry blocks)
    h->arena_recs = 2;
    ok &= dm((void**)&h->arena, (size_t)h->arena_recs * h->rec_stride * 4);
    ok &= hipHostMalloc((void**)&h->h_ctl, sizeof(StepState)) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_meta, (size_t)(c->max_attempts + 1) * sizeof(StepMeta)) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_init, sizeof(InitRec)) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_scal, 64 * sizeof(float)) == hipSuccess;
    if (!ok) { g_create_err = "device allocation failed"; rnde_node_destroy(h); return RNDE_ERR_HIP; }
    hipMemset(h->initrec, 0, sizeof(InitRec));
    h->predicted = 12;
    *out = h;
    return RNDE_OK;
}
static ChainParams make_chain_params(rnde_node* h, const StepParams& P) {
    ChainParams Q{};
    Q.F = P; Q.G = h->cg; Q.frags = h->cfrags; Q.ntiles = P.Bpad / 16;
    return Q;
}
template <int NKD, int MODE, int ALT = 0>
static hipError_t launch_chain_t(rnde_node* h, const ChainParams& Q, int n, float* u_out, hipStream_t s) {
    auto kern = rnde_chain_kernel<NKD, MODE, ALT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(Q.F.nwg), dim3(64 * kCW), MODE == CM_FINISH ? 0 : h->chain_lds_f, s, Q, n, u_out);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_chain(rnde_node* h, const ChainParams& Q, int n, float* u_out, hipStream_t s) {
    switch (h->NKD) {
        case 4: return launch_chain_t<4, MODE>(h, Q, n, u_out, s);
        case 8: return h->chain_alt ? launch_chain_t<8, MODE, 1>(h, Q, n, u_out, s) : launch_chain_t<8, MODE>(h, Q, n, u_out, s);
        default: return launch_chain_t<16, MODE>(h, Q, n, u_out, s);
    }
}
static MwParams make_mw_params(rnde_node* h, const StepParams& P) {
    MwParams Q{};
    Q.F = P; Q.G = h->mg; Q.rk = h->rk; Q.tab = h->mw_tab; Q.ntiles = P.Bpad / 16; Q.u_out = nullptr;
    Q.ev_stride = (long long)Q.ntiles * h->mg.RS * 64;
    Q.slab = P.tape ? h->mw_slab : nullptr;
    return Q;
}
template <int NR, int MODE, int TAB, int LAT = 0>
static hipError_t launch_mw_t(rnde_node* h, const MwParams& Q, int n, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_chainmw_kernel<NR, MODE, TAB, LAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((rnde_chainmw_kernel<NR, MODE, TAB, LAT>), dim3((MODE == MW_SOLVE && !Q.xch_global) ? 8 * Q.ntiles : Q.ntiles), dim3(kMwThreads), MODE == MW_FINISH ? 0 : h->mw_lds_f, s, Q, n);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_mw(rnde_node* h, const MwParams& Q, int n, hipStream_t s) {
    if (h->rk_tab == 2) {      // S-stage table
        switch (h->NKD) {
            case 4: return launch_mw_t<1, MODE, 2>(h, Q, n, s);
            case 8: return launch_mw_t<2, MODE, 2>(h, Q, n, s);
            default: return launch_mw_t<4, MODE, 2>(h, Q, n, s);
        }
    }
    if (h->mw_lat) return h->rk_tab ? launch_mw_t<2, MODE, 1, 1>(h, Q, n, s) : launch_mw_t<2, MODE, 0, 1>(h, Q, n, s);   // latent-ODE shape: weights register stationary
    if (h->rk_tab) {
        switch (h->NKD) {
            case 4: return launch_mw_t<1, MODE, 1>(h, Q, n, s);
            case 8: return launch_mw_t<2, MODE, 1>(h, Q, n, s);
            default: return launch_mw_t<4, MODE, 1>(h, Q, n, s);
        }
    }
    switch (h->NKD) {
        case 4: return launch_mw_t<1, MODE, 0>(h, Q, n, s);
        case 8: return launch_mw_t<2, MODE, 0>(h, Q, n, s);
        default: return launch_mw_t<4, MODE, 0>(h, Q, n, s);
    }
}
// the forward tapes every layer input of every evaluation into the slab: make room for `evals` evaluations (growing keeps what is there)
static rnde_status ensure_mw_slab(rnde_node* h, long long evals, int Bpad, hipStream_t s) {
    const long long per_eval = (long long)(h->Bpad_max / 16) * h->mg.RS * 64;   // sized for max_batch so that ev_stride changes never outgrow it
    (void)Bpad;
    if (h->mw_slab_evals >= evals) return RNDE_OK;
    const long long want = std::max(evals, 2 * h->mw_slab_evals);
    float* nb = nullptr;
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipMalloc((void**)&nb, (size_t)want * per_eval * 4));
    if (h->mw_slab) {
        HIPCHK(h, hipMemcpy(nb, h->mw_slab, (size_t)h->mw_slab_evals * per_eval * 4, hipMemcpyDeviceToDevice));
        hipFree(h->mw_slab);
    }
    h->mw_slab = nb; h->mw_slab_evals = want;
    return RNDE_OK;
}

static rnde_status chain_pack(rnde_node* h, const float* p_dev, hipStream_t s) {
    if (h->mw) {
        const long long tm = (long long)mw_tab_floats(h->mg);
        hipLaunchKernelGGL(rnde_chainmw_pack_kernel, dim3((int)std::min<long long>((tm + 255) / 256, 512)), dim3(256), 0, s, p_dev, h->mw_tab, h->mg);
        HIPCHK(h, hipGetLastError());
        return RNDE_OK;
    }
    const long long total = (long long)(h->cg.nfrag_f + h->cg.nfrag_b + h->cg.nfrag_t) * 64;
    hipLaunchKernelGGL(rnde_chain_pack_kernel, dim3((int)std::min<long long>((total + 255) / 256, 512)), dim3(256), 0, s, p_dev, h->cfrags, h->cg);
    HIPCHK(h, hipGetLastError());
    return RNDE_OK;
}
static hipError_t chain_convert(const float* src, float* dst, int D, int B, int ntiles, int nksD, int to_caller, hipStream_t s) {
    const long long total = (long long)ntiles * nksD * 64;
    hipLaunchKernelGGL(rnde_chain_convert_kernel, dim3((int)std::min<long long>((total + 255) / 256, 512)), dim3(256), 0, s, src, dst, D, B, ntiles, nksD, to_caller);
    return hipGetLastError();
}

extern "C" rnde_status rnde_node_create(const rnde_node_config* c, rnde_node** out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c->device) { g_create_err = "no HIP device"; return RNDE_ERR_NO_DEVICE; }
    if ((c->solver != RNDE_SOLVER_TSIT5 && c->solver != RNDE_SOLVER_DP5 && c->solver != RNDE_SOLVER_DOP853) || c->n_layers < 1 || c->n_layers > RNDE_MAX_LAYERS || c->dims[0] != c->dims[c->n_layers]) {
        g_create_err = "unsupported configuration: Tsit5 (or DP5) over a Dense chain with dims[0] == dims[n_layers]"; return RNDE_ERR_BAD_ARG;
    }
    const bool mnist_form = c->n_layers == 2 && c->time_dep && !c->pre_act && c->act[0] == RNDE_ACT_TANH;
    if ((c->solver == RNDE_SOLVER_DP5 || c->solver == RNDE_SOLVER_DOP853) && ((mnist_form && c->col_tile != 65) || c->col_tile == 64 || c->regularize >= RNDE_REG_STIFF)) {
        g_create_err = "DP5 / DOP853 run on the tableau-as-data kernels of the chain engine (col_tile 0 for Dense chains of width <= 64, or 65), callbacks none / EEst*dt";
        return RNDE_ERR_BAD_ARG;
    }
    if (c->col_tile == 64 || c->col_tile == 65 || !mnist_form) return chain_create(c, out);   // small-width chains (latent_ode.jl:113-124): rnde_chain.h
    if (c->regularize < RNDE_REG_NONE || c->regularize > RNDE_REG_ERR_STIFF) { g_create_err = "regularize: unknown value"; return RNDE_ERR_BAD_ARG; }
    rnde_node* h = new rnde_node();
    h->cfg = *c;
    h->D = c->dims[0]; h->H = c->dims[1]; h->P = rnde_param_count(c); h->act2 = c->act[1];
    h->BT = 8;   // (column granularity of the batch padding the tape's copy of x is zero-filled to)
    if (c->col_tile != 0 && c->col_tile != 16) {
        g_create_err = "col_tile must be 0 (auto) or 16 (stage engine) for the two-layer TDChain form; 64 / 65 select the chain engine.  (The round-1 column-owner "
                       "engine, col_tile 4 / 8, was retired in round 4: 126 us per attempted step against 24; its sources are kept under tools/experiments/column_owner/.)";
        delete h; return RNDE_ERR_BAD_ARG;
    }
    if (h->H + 2 > 128 || h->D < 1 || h->H < 1 || (h->D + 15) / 16 > 64 || c->max_batch < 1 || c->max_attempts < 1) {
        g_create_err = "shape outside the kernel limits (H <= 126, D <= 1024)"; delete h; return RNDE_ERR_BAD_ARG;
    }
    h->Bpad_max = ((c->max_batch + 15) / 16) * 16;
    // stage engine geometry: WT row tiles (waves) per block chosen to minimise padding, preferring more waves
    h->sMT = (h->D + 15) / 16; h->sHT = (h->H + 1 + 15) / 16; h->sK2b = (h->H + 2 + 15) / 16; h->sKHb = (h->H + 15) / 16;
    { int best = 1, bw = 1 << 30;
      for (int wt = std::min(8, h->sMT); wt >= std::max(1, std::min(4, h->sMT)); --wt) { int R = (h->sMT + wt - 1) / wt; int waste = R * wt - h->sMT; if (waste < bw) { bw = waste; best = wt; } }
      if (const char* e = getenv("RNDE_STAGE_WT")) { const int v = atoi(e); if (v >= 1 && v <= 8) best = std::min(v, h->sMT); }   // (experiments: row tiles per workgroup)
      h->sWT = best; h->sR = (h->sMT + best - 1) / best; }
    h->engine = 2;
    h->nwg_max = std::max(h->Bpad_max / h->BT, h->sR * (h->Bpad_max / 16));
    h->stage_lds = sizeof(float) * ((size_t)16 * (16 * std::max(h->sK2b, h->sHT) + 4) + (size_t)16 * (16 * std::max(h->sWT, h->sKHb) + 4) + 64);
    if (hipSetDevice(c->device) != hipSuccess) { g_create_err = "hipSetDevice failed"; delete h; return RNDE_ERR_HIP; }
    const size_t A = (size_t)h->D * h->Bpad_max, HB = (size_t)h->H * h->Bpad_max;
    RecLayout L{(long long)A, (long long)HB};
    h->rec_stride = L.total();
    auto dm = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    bool ok = true;
    ok &= dm((void**)&h->f0, A * 4) && dm((void**)&h->u1, A * 4) && dm((void**)&h->f1, A * 4) && dm((void**)&h->xcopy, A * 4);
    ok &= dm((void**)&h->h0, HB * 4) && dm((void**)&h->h1, HB * 4);
    ok &= dm((void**)&h->pcopy, (size_t)h->P * 4);
    ok &= dm((void**)&h->spwB, (size_t)h->sMT * h->sK2b * 64 * 16) && dm((void**)&h->spwD, (size_t)h->sHT * h->sMT * 64 * 16);
    ok &= dm((void**)&h->spwBt, (size_t)h->sMT * h->sKHb * 64 * 16) && dm((void**)&h->spwDt, (size_t)h->sHT * h->sMT * 64 * 16);
    ok &= dm((void**)&h->slab2, (size_t)2 * (h->Bpad_max / 16) * h->sR * h->sHT * 64 * 16);
    const size_t tslab_bytes = (size_t)3 * (h->Bpad_max / 16) * h->sR * h->sHT * 64 * 16;    // three buffers of one 16-byte entry per lane and tile
    h->tslab_bytes = tslab_bytes;
    ok &= dm((void**)&h->tslab, tslab_bytes);
    // mailbox: [0) final controller state | [512) initial-step record | [1016) abort word of the persistent kernels, [1024) their
    // XCC ids | [meta_off) step metadata -- read by the host with ONE copy per chunk (was five)
    static_assert(sizeof(StepState) <= 512 && sizeof(InitRec) <= 504, "mailbox layout");
    h->mbox_meta_off = (1024 + (size_t)h->nwg_max * 4 + 255) / 256 * 256;
    const size_t mbox_bytes = h->mbox_meta_off + (size_t)(c->max_attempts + 1) * sizeof(StepMeta);
    ok &= dm((void**)&h->mbox, mbox_bytes) && hipHostMalloc((void**)&h->h_mbox, mbox_bytes) == hipSuccess;
    if (ok) {
        hipMemset(h->mbox, 0, mbox_bytes);
        h->ctl_final = (StepState*)h->mbox; h->initrec = (InitRec*)(h->mbox + 512);
        h->pabort = (unsigned*)(h->mbox + 1016); h->pxcc = (unsigned*)(h->mbox + 1024); h->meta = (StepMeta*)(h->mbox + h->mbox_meta_off);
        h->h_ctl = (StepState*)h->h_mbox; h->h_init = (InitRec*)(h->h_mbox + 512);
        h->h_pchk = (unsigned*)(h->h_mbox + 1016); h->h_meta = (StepMeta*)(h->h_mbox + h->mbox_meta_off);
    }
    ok &= dm((void**)&h->ctl, 2 * sizeof(StepState));
    ok &= dm((void**)&h->errpart, (size_t)(6 * h->nwg_max + 256) * 4) && dm((void**)&h->initpart, (size_t)(3 * h->nwg_max + 256) * 4);   // (+256: sum_partials reads whole 256-entry blocks)
    // scratch records: 2 (no-tape ring); grown to max_attempts on the first taped forward
    h->arena_recs = 2;
    ok &= dm((void**)&h->arena, (size_t)h->arena_recs * h->rec_stride * 4);
    ok &= hipHostMalloc((void**)&h->h_scal, 64 * sizeof(float)) == hipSuccess;
    if (!ok) { g_create_err = "device allocation failed"; rnde_node_destroy(h); return RNDE_ERR_HIP; }
    hipMemset(h->tslab, 0xFF, tslab_bytes); hipMemset(h->pabort, 0, 8); hipMemset(h->pxcc, 0, (size_t)h->nwg_max * 4);
    {   // tuning knobs: config fields, each with a create-time environment override for A/B tooling (never read per solve)
        const char* e = getenv("RNDE_PERSIST");
        const bool off = c->persist < 0 || (e && e[0] == '0');
        h->persist = (h->engine == 2 && h->sR <= 8 && !off) ? 1 : 0;
        if (const char* e2 = getenv("RNDE_PERSIST_SPINS")) h->persist_spins = atoi(e2);
        if (const char* e3 = getenv("RNDE_PERSIST2")) h->persist2 = atoi(e3);
        h->wgrad_side_pct = c->wgrad_side_pct < 0 ? 0 : (c->wgrad_side_pct == 0 ? 30 : std::min(100, c->wgrad_side_pct));
        if (const char* e3 = getenv("RNDE_WGRAD_SIDE")) h->wgrad_side_pct = atoi(e3);
        h->stage_generic = (c->stage_generic != 0 || getenv("RNDE_STAGE_GENERIC") != nullptr) ? 1 : 0;
        if (const char* e6 = getenv("RNDE_STAGE_SOLVE")) h->stage_solve = atoi(e6);
    }
    if (h->persist == 1 && h->stage_solve && c->max_attempts < 8192) {      // meeting granules of the one-launch solve: [attempt][3][256] x 8 bytes
        const size_t xb = (size_t)(c->max_attempts + 1) * 3 * 256 * 8;
        if (hipMalloc((void**)&h->sxch, xb) != hipSuccess) { g_create_err = "device allocation failed"; rnde_node_destroy(h); return RNDE_ERR_HIP; }
        hipMemset(h->sxch, 0, xb);
    }
    h->predicted = 12;
    *out = h;
    return RNDE_OK;
}

extern "C" void rnde_node_destroy(rnde_node* h) {
    if (!h) return;
    void* d[] = {h->f0, h->h0, h->u1, h->f1, h->h1, h->arena, h->xcopy, h->pcopy, h->spwB, h->spwD, h->spwBt, h->spwDt, h->slab2,
                 h->ctl, h->ctl_final, h->meta, h->initrec, h->errpart, h->initpart};
    if (h->mbox || h->h_mbox) {   // these alias the mailbox
        for (void*& p : d) if (p == h->ctl_final || p == h->meta || p == h->initrec) p = nullptr;
        h->pabort = h->pxcc = nullptr; h->h_pchk = nullptr; h->h_ctl = nullptr; h->h_meta = nullptr; h->h_init = nullptr;
        if (h->mbox) hipFree(h->mbox);
        if (h->h_mbox) hipHostFree(h->h_mbox);
    }
    for (void* p : d) if (p) hipFree(p);
    bwd_free(h->bw);
    if (h->head_ws) hipFree(h->head_ws);
    if (h->cg_ws) hipFree(h->cg_ws);
    if (h->ev_host) hipEventDestroy(h->ev_host);
    if (h->sv_t_dev) hipFree(h->sv_t_dev);
    if (h->replay_dev) hipFree(h->replay_dev);
    if (h->cfrags) hipFree(h->cfrags);
    if (h->mw_tab) hipFree(h->mw_tab);
    if (h->mw_xch) hipFree(h->mw_xch);
    if (h->mw_xcc) hipFree(h->mw_xcc);
    if (h->mw_abort) hipFree(h->mw_abort);
    if (h->h_mw_chk) hipHostFree(h->h_mw_chk);
    if (h->h_mw_bchk) hipHostFree(h->h_mw_bchk);
    if (h->h_mw_bargs) hipHostFree(h->h_mw_bargs);
    if (h->mw_bargs) hipFree(h->mw_bargs);
    if (h->mw_slab) hipFree(h->mw_slab);
    if (h->tslab) hipFree(h->tslab);
    if (h->sxch) hipFree(h->sxch);
    if (h->pabort) hipFree(h->pabort);
    if (h->pxcc) hipFree(h->pxcc);
    if (h->h_pchk) hipHostFree(h->h_pchk);
    if (h->cslab) hipFree(h->cslab);
    if (h->ev_t) hipFree(h->ev_t);
    if (h->h_ev_t) hipHostFree(h->h_ev_t);
    for (hipEvent_t e : h->wevents) hipEventDestroy(e);
    for (hipEvent_t e : h->tev) if (e) hipEventDestroy(e);
    if (h->wstream) hipStreamDestroy(h->wstream);
    if (h->h_ctl) hipHostFree(h->h_ctl);
    if (h->h_meta) hipHostFree(h->h_meta);
    if (h->h_init) hipHostFree(h->h_init);
    if (h->h_scal) hipHostFree(h->h_scal);
    delete h;
}

// ---- SURVEY 8e mode 2: one controller for all shards (include/rnde.h: rnde_node_set_coupling).  Every launch that leaves per-workgroup
// partial sums of a batch-wide norm is followed by an all-reduce of that partial array over the shards, on the same stream: the kernels
// that consume the partials are unchanged (they sum the array in a fixed order, which now holds the element-wise sums over the ranks).
static rnde_status couple_sum(rnde_node* h, float* partials, long long count, hipStream_t s) {
    if (!h->couple) return RNDE_OK;
    const rnde_status st = rnde_comm_allreduce(h->couple, partials, count, 0, (void*)s);
    if (st != RNDE_OK) h->err = std::string("coupled controller: ") + rnde_comm_last_error(h->couple);
    return st;
}
extern "C" rnde_status rnde_node_set_coupling(rnde_node* h, rnde_comm* c, int32_t global_batch) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (!c) { h->couple = nullptr; h->couple_batch = 0; h->couple_world = 1; return RNDE_OK; }
    const int world = rnde_comm_world(c);
    if (h->engine != 2 && !(h->engine == 3 && h->mw)) { h->err = "coupled controller: the stage engine (MNIST form, col_tile 0 or 16) and the chain engine's multi-wave kernels only"; return RNDE_ERR_BAD_ARG; }
    if (global_batch < world || world < 1) { h->err = "coupled controller: global_batch must cover every rank"; return RNDE_ERR_BAD_ARG; }
    h->couple = c; h->couple_batch = global_batch; h->couple_world = world;
    return RNDE_OK;
}

static rnde_status ensure_arena(rnde_node* h, long long recs) {
    if (h->arena_recs >= recs) return RNDE_OK;
    if (h->arena) hipFree(h->arena);
    h->arena = nullptr; h->arena_recs = 0;
    HIPCHK(h, hipMalloc((void**)&h->arena, (size_t)recs * h->rec_stride * 4));
    h->arena_recs = recs;
    return RNDE_OK;
}

static StageParams make_stage_params(rnde_node* h, const StepParams& P, const float* p_dev) {
    StageParams Q{};
    Q.F = P; Q.p = p_dev; Q.pwB = h->spwB; Q.pwD = h->spwD; Q.slab = h->slab2;
    Q.MT = h->sMT; Q.WT = h->sWT; Q.R = h->sR; Q.C = P.Bpad / 16; Q.HT = h->sHT; Q.K2b = h->sK2b; Q.Bpad16 = P.Bpad;
    return Q;
}
template <int ACT2, int MODE>
static hipError_t launch_stage_t(rnde_node* h, const StageParams& Q, int n, int s, hipStream_t st) {
    hipLaunchKernelGGL((rnde_stage_kernel<ACT2, MODE>), dim3(Q.R * Q.C), dim3(64 * Q.WT), h->stage_lds, st, Q, n, s);
    return hipGetLastError();
}
template <int MODE>
static hipError_t launch_stage(rnde_node* h, const StageParams& Q, int n, int s, hipStream_t st) {
    return h->act2 ? launch_stage_t<1, MODE>(h, Q, n, s, st) : launch_stage_t<0, MODE>(h, Q, n, s, st);
}
static hipError_t stage_pack(rnde_node* h, const float* p, f32x4* dst, int which, int MTrows, int Kb, hipStream_t st) {
    const long long total = (long long)MTrows * Kb * 64;
    const int grid = (int)std::min<long long>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(rnde_stage_pack_kernel, dim3(grid), dim3(256), 0, st, p, dst, which, h->D, h->H, MTrows, Kb);
    return hipGetLastError();
}
static rnde_status stage_pack_weights(rnde_node* h, const float* p_dev, hipStream_t s) {
    HIPCHK(h, stage_pack(h, p_dev, h->spwB, 0, h->sMT, h->sK2b, s));
    HIPCHK(h, stage_pack(h, p_dev, h->spwD, 1, h->sHT, h->sMT, s));
    return RNDE_OK;
}
static rnde_status stage_pack_all(rnde_node* h, const float* p_dev, hipStream_t s, const float* x_src = nullptr, long long x_floats = 0) {
    PackJobs J{};
    int n = 0;
    auto add = [&](void* dst, long long total, int kind, int which, int kdim, const float* src = nullptr) { J.j[n++] = PackJob{dst, src, total, kind, which, kdim, 0}; };
    add(h->spwB, (long long)h->sMT * h->sK2b * 64, 0, 0, h->sK2b);
    add(h->spwD, (long long)h->sHT * h->sMT * 64, 0, 1, h->sMT);
    add(h->spwBt, (long long)h->sMT * h->sKHb * 64, 0, 2, h->sKHb);
    add(h->spwDt, (long long)h->sHT * h->sMT * 64, 0, 3, h->sMT);
    auto add_copy = [&](float* dst, const float* src, long long floats) {     // 16-byte copies where sizes and addresses allow
        const bool v4 = floats % 4 == 0 && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)(src ? src : p_dev) % 16 == 0);
        add(dst, v4 ? floats / 4 : floats, v4 ? 3 : 2, 0, 0, src);
    };
    add_copy(h->pcopy, nullptr, (long long)h->P);
    if (x_src) add_copy(h->xcopy, x_src, x_floats);           // the tape's copy of x rides along (was a launch of its own)
    long long most = 0;
    for (int i = 0; i < n; ++i) most = std::max(most, J.j[i].total);
    const int grid = (int)std::min<long long>((most + 255) / 256, 256);
    hipLaunchKernelGGL(rnde_pack_all_kernel, dim3(grid, n), dim3(256), 0, s, p_dev, J, h->D, h->H);
    HIPCHK(h, hipGetLastError());
    h->rev_packed = true;
    return RNDE_OK;
}
// The hand-off slabs must read "empty" wherever the persistent kernels have not written in the current tile indexing: at
// creation, and whenever the padded batch width (= number of column tiles) differs from the last persistent launch's.
static hipError_t slab_prepare(rnde_node* h, int Bpad, hipStream_t s) {
    if (h->tslab_Bpad == Bpad) return hipSuccess;
    h->tslab_Bpad = Bpad;
    return hipMemsetAsync(h->tslab, 0xFF, h->tslab_bytes, s);
}
static hipError_t stage_attempt(rnde_node* h, const StageParams& Q, int n, hipStream_t s) {
    if (h->persist == 1) {   // one launch per attempt, slab hand-offs inside the kernel (rnde_stage_persist.h)
        PersistSync Y{h->tslab, h->pabort, h->pxcc, h->persist_spins};
        if (hipError_t e = slab_prepare(h, Q.Bpad16, s); e != hipSuccess) return e;
        const dim3 grid(8 * Q.R * ((Q.C + 7) / 8));   // a column tile's row blocks share blockIdx % 8 (same XCD)
        const bool fix = Q.WT == 7 && Q.HT == 7 && Q.K2b == 7 && Q.MT == 49 && Q.R == 7 && h->D == 784 && h->H == 100 && !h->stage_generic;
        // batches that fill the chip more than once: two column tiles per workgroup (rnde_stage_persist2.h; bit-identical results).
        // RNDE_PERSIST2=0 keeps one tile per workgroup (A/B and the bit-identity test), =1 takes two whenever the tile count is even.
        if (fix && Q.C % 2 == 0 && h->persist2 != 0 && (Q.C >= kPersist2MinTiles || h->persist2 >= 1)) {
            const dim3 grid2(8 * Q.R * ((Q.C / 2 + 7) / 8));
            const size_t lds2 = sizeof(float) * (2 * 2 * 16 * (16 * 7 + 4) + 32 * 3);
            if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<1, 2>), grid2, dim3(64 * 7), lds2, s, Q, n, Y);
            else hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<0, 2>), grid2, dim3(64 * 7), lds2, s, Q, n, Y);
            return hipGetLastError();
        }
        if (fix) {
            if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_kernel<1, 1>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y);
            else hipLaunchKernelGGL((rnde_stage_attempt_kernel<0, 1>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y);
        } else if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_kernel<1, 0>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y);
        else hipLaunchKernelGGL((rnde_stage_attempt_kernel<0, 0>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y);
        return hipGetLastError();
    }
    hipError_t e = launch_stage<SM_START>(h, Q, n, 0, s);
    for (int st = 1; st <= 5 && e == hipSuccess; ++st) e = launch_stage<SM_STAGE>(h, Q, n, st, s);
    if (e == hipSuccess) e = launch_stage<SM_LAST>(h, Q, n, 6, s);
    return e;
}

// The one-launch reverse sweep of the chain engine (rnde_bchainmw.h SWEEP) leaves its verdict in h_mw_bchk behind the sweep; looked at after the
// next stream synchronisation.  true: a meeting timed out or the workgroups did not share an XCD -- the sweep's outputs are invalid, the handle
// goes back to one launch per reversed attempt.
static bool bsweep_failed(rnde_node* h, hipStream_t s) {
    if (!h->pending_bsweep) return false;
    h->pending_bsweep = false;
    bool bad = h->h_mw_bchk[0] != 0;
    const int nt = h->bw.ready ? (int)((h->B + 15) / 16) : 0;
    for (int i = 1; i < nt && nt <= 32 && !bad; ++i) bad = h->h_mw_bchk[2 + i] != h->h_mw_bchk[2];      // (more than 32 tiles: the meeting does not depend on the placement)
    if (!bad) return false;
    fprintf(stderr, "[rnde] chain engine: one-launch reverse sweep abandoned (%s); one launch per reversed attempt for the next %d solves\n",
            h->h_mw_bchk[0] ? "a meeting timed out" : "workgroups pinned by block index landed on different XCDs", h->mw_retry_after);
    h->mw_bsweep = -1; h->mw_clean = 0; ++h->persist_fallbacks;
    hipMemsetAsync(h->mw_abort, 0, 8, s);
    h->h_mw_bchk[0] = 0;
    return true;
}

static rnde_status forward_core(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                float* u_out_dev, const float* saveat_host, int32_t n_saveat, float* sv_out_dev,
                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream);
static const rnde_status RNDE_INTERNAL_RETRY = static_cast<rnde_status>(100);

// After a synchronisation point: did a persistent launch time out, or did a column tile's workgroups land on different
// XCDs (then their slab hand-off through L2 would not be coherent)?  Either way the persistent kernels are disabled for
// this handle and the caller redoes the solve with the multi-launch kernels.
// (two halves so that the two small copies ride on a synchronisation the caller performs anyway)
static void persist_check_enqueue(rnde_node* h, int grid, hipStream_t s) {
    if (h->persist != 1) return;
    h->h_pchk[0] = 1;   // stays 1 ("failed") if a copy cannot even be enqueued
    if (hipMemcpyAsync(h->h_pchk, h->pabort, 8, hipMemcpyDeviceToHost, s) != hipSuccess) return;
    (void)hipMemcpyAsync(h->h_pchk + 2, h->pxcc, (size_t)grid * 4, hipMemcpyDeviceToHost, s);
}
static bool persist_check_result(rnde_node* h, int C, int R, hipStream_t s) {   // call after the stream has been synchronised
    if (h->persist != 1) return false;
    bool bad = h->h_pchk[0] != 0;
    for (int ct = 0; ct < C && !bad; ++ct)
        for (int rb = 1; rb < R; ++rb) if (h->h_pchk[2 + rb * C + ct] != h->h_pchk[2 + ct]) { bad = true; break; }
    if (bad) {
        fprintf(stderr, "[rnde] persistent attempt kernel suspended (%s); using the multi-launch kernels for the next %d solves\n",
                h->h_pchk[0] ? "hand-off timed out" : "column tile spans XCDs", h->persist_retry_after);
        h->persist = -1; h->persist_clean = 0; ++h->persist_fallbacks;
        hipMemsetAsync(h->pabort, 0, 8, s);
    }
    return bad;
}
static bool persist_failed(rnde_node* h, int grid, int C, int R, hipStream_t s) {
    if (h->persist != 1) return false;
    persist_check_enqueue(h, grid, s);
    if (hipStreamSynchronize(s) != hipSuccess) return true;
    return persist_check_result(h, C, R, s);
}

static rnde_status forward_impl(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                float* u_out_dev, const float* saveat_host, int32_t n_saveat, float* sv_out_dev,
                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    rnde_status st = forward_core(h, x_dev, p_dev, B, t0, t1, u_out_dev, saveat_host, n_saveat, sv_out_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    if (st == RNDE_INTERNAL_RETRY)
        st = forward_core(h, x_dev, p_dev, B, t0, t1, u_out_dev, saveat_host, n_saveat, sv_out_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    return st;
}

extern "C" rnde_status rnde_node_forward(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0,
                                         float t1, float* u_out_dev, int64_t* nfe_out, float* saveval_host,
                                         int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    return forward_impl(h, x_dev, p_dev, B, t0, t1, u_out_dev, nullptr, 0, nullptr, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
}

extern "C" rnde_status rnde_node_forward_replay(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0,
                                                float t1, const float* steps_host, int32_t n_steps, float* u_out_dev,
                                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out,
                                                int32_t keep_tape, void* stream) {
    if (!h || !steps_host || n_steps < 1) return RNDE_ERR_BAD_ARG;
    if (n_steps > h->cfg.max_attempts) { h->err = "replay: more steps than max_attempts"; return RNDE_ERR_BAD_ARG; }
    h->replay_host = steps_host; h->n_replay = n_steps;
    const rnde_status st = forward_impl(h, x_dev, p_dev, B, t0, t1, u_out_dev, nullptr, 0, nullptr, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    h->replay_host = nullptr; h->n_replay = 0;
    return st;
}

extern "C" rnde_status rnde_node_forward_saveat(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0,
                                                float t1, const float* saveat_host, int32_t n_saveat, float* u_saved_dev,
                                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out,
                                                int32_t keep_tape, void* stream) {
    if (!h || !saveat_host || n_saveat < 1 || !u_saved_dev) return RNDE_ERR_BAD_ARG;
    return forward_impl(h, x_dev, p_dev, B, t0, t1, nullptr, saveat_host, n_saveat, u_saved_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
}

static rnde_status forward_core(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                float* u_out_dev, const float* saveat_host, int32_t n_saveat, float* sv_out_dev,
                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    if (!h) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_saveat > 0) {
        if (h->rk_tab == 2) { h->err = "saveat: this Runge-Kutta table carries no dense output (DOP853)"; return RNDE_ERR_BAD_ARG; }
        for (int i = 0; i < n_saveat; ++i)
            if (!(saveat_host[i] >= t0 && saveat_host[i] <= t1) || (i > 0 && !(saveat_host[i] > saveat_host[i - 1]))) {
                h->err = "saveat must be increasing and inside [t0, t1]"; return RNDE_ERR_BAD_ARG;
            }
        if ((size_t)n_saveat > h->sv_cap) {
            if (h->sv_t_dev) hipFree(h->sv_t_dev);
            h->sv_t_dev = nullptr; h->sv_cap = 0;
            HIPCHK(h, hipMalloc((void**)&h->sv_t_dev, (size_t)n_saveat * 4));
            h->sv_cap = n_saveat;
        }
        h->saveat.assign(saveat_host, saveat_host + n_saveat);
        HIPCHK(h, hipMemcpyAsync(h->sv_t_dev, h->saveat.data(), (size_t)n_saveat * 4, hipMemcpyHostToDevice, s));
    } else h->saveat.clear();
    if (B < 1 || B > h->cfg.max_batch || !(t1 > t0)) { h->err = "bad B or tspan"; return RNDE_ERR_BAD_ARG; }
    // the coupled controller all-reduces per-workgroup partial arrays element by element: every rank must hold the same number of columns
    if (h->couple && (long long)B * h->couple_world != h->couple_batch) {
        h->err = "coupled controller: equal shards only (B * world must equal the global batch given to rnde_node_set_coupling)";
        return RNDE_ERR_BAD_ARG;
    }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->have_tape = false; h->rev_packed = false;
    const float* x_caller = nullptr;
    if (keep_tape) {
        rnde_status st = ensure_arena(h, h->cfg.max_attempts);
        if (st != RNDE_OK) return st;
        // the tape owns copies of x and p (the caller may free or overwrite its buffers before backward)
        if (B % h->BT) HIPCHK(h, hipMemsetAsync(h->xcopy, 0, (size_t)h->D * (((B + h->BT - 1) / h->BT) * h->BT) * 4, s));
        if (h->engine != 2) {   // (stage engine: both copies are part of the one pack launch below)
            HIPCHK(h, hipMemcpyAsync(h->xcopy, x_dev, (size_t)h->D * B * 4, hipMemcpyDeviceToDevice, s));
            HIPCHK(h, hipMemcpyAsync(h->pcopy, p_dev, (size_t)h->P * 4, hipMemcpyDeviceToDevice, s));
        } else x_caller = x_dev;
        x_dev = h->xcopy;
    }
    StepParams P = make_params(h, x_dev, B, t0, t1, keep_tape ? 1 : 0);
    P.sv_t = n_saveat > 0 ? h->sv_t_dev : nullptr; P.nsave = n_saveat; P.sv_out = sv_out_dev;
    if (h->n_replay > 0) {
        if ((size_t)h->n_replay > h->replay_cap) {
            if (h->replay_dev) hipFree(h->replay_dev);
            h->replay_dev = nullptr; h->replay_cap = 0;
            HIPCHK(h, hipMalloc((void**)&h->replay_dev, (size_t)h->n_replay * 8));
            h->replay_cap = h->n_replay;
        }
        HIPCHK(h, hipMemcpyAsync(h->replay_dev, h->replay_host, (size_t)h->n_replay * 8, hipMemcpyHostToDevice, s));
        P.replay = h->replay_dev; P.n_replay = h->n_replay;
    }
    h->B = B; h->Bpad = P.Bpad; h->nwg = P.nwg; h->t0 = t0; h->t1 = t1;
    rnde_status st = RNDE_OK;
    if (h->engine == 2 && keep_tape) st = stage_pack_all(h, p_dev, s, x_caller, (long long)h->D * B);   // forward + reverse packs of both engines' layouts and the tape's copy of p: one launch
    else if (h->engine == 3) st = chain_pack(h, keep_tape ? h->pcopy : p_dev, s);
    if (st != RNDE_OK) return st;
    StageParams SQ{};
    ChainParams CQ{};
    MwParams MQ{};
    if (h->engine == 3) {
        CQ = make_chain_params(h, P);
        if (h->mw) {
            if (keep_tape) { st = ensure_mw_slab(h, 2 + (long long)(h->rk_S - 1) * std::max(4, h->predicted), P.Bpad, s); if (st != RNDE_OK) return st; }
            MQ = make_mw_params(h, P);
            HIPCHK(h, launch_mw<MW_INIT_A>(h, MQ, 0, s));
            if ((st = couple_sum(h, P.initpart, 2LL * P.nwg, s)) != RNDE_OK) return st;            // (coupled controller: norms of u0 and f0)
            HIPCHK(h, launch_mw<MW_INIT_B>(h, MQ, 0, s));
            if ((st = couple_sum(h, P.initpart + 2LL * P.nwg, P.nwg, s)) != RNDE_OK) return st;     // norm of f1 - f0
        } else {
            HIPCHK(h, launch_chain<CM_INIT_A>(h, CQ, 0, nullptr, s));
            HIPCHK(h, launch_chain<CM_INIT_B>(h, CQ, 0, nullptr, s));
        }
    } else if (h->engine == 2) {
        if (!keep_tape) { st = stage_pack_weights(h, p_dev, s); if (st != RNDE_OK) return st; }
        SQ = make_stage_params(h, P, keep_tape ? h->pcopy : p_dev);
        HIPCHK(h, launch_stage<SM_I1>(h, SQ, 0, 0, s));
        HIPCHK(h, launch_stage<SM_I2>(h, SQ, 0, 0, s));
        if ((st = couple_sum(h, P.initpart, 2LL * P.nwg, s)) != RNDE_OK) return st;            // norms of u0 and f0
        HIPCHK(h, launch_stage<SM_I3>(h, SQ, 0, 0, s));
        HIPCHK(h, launch_stage<SM_I4>(h, SQ, 0, 0, s));
        if ((st = couple_sum(h, P.initpart + 2LL * P.nwg, P.nwg, s)) != RNDE_OK) return st;     // norm of f1 - f0
    }
    int launched = 0;
    int chunk = h->couple ? 16 : std::max(4, h->predicted);   // (coupled: the same launch count on every rank, whatever its history)
    const int cap = h->cfg.max_attempts;
    h->tev_fwd = false;
    if (h->timing) HIPCHK(h, hipEventRecord(h->tev[0], s));
    // ---- chain engine, multi-wave kernels, every workgroup resident (<= 256 column tiles): the WHOLE adaptive solve is one launch (attempt loop,
    // ---- controller and the once-per-attempt meeting of the workgroups inside the kernel).  <= 32 tiles: the workgroups are pinned to one XCD
    // ---- and meet through its L2; more (B > 512, the throughput case): they spread over the chip and meet through agent-scope entries ----
    bool solved = false;
    if (h->engine == 3 && h->mw && h->mw_solve > 0 && !h->couple && P.Bpad / 16 <= kMwMeetMax) {
        // a taped solve writes every layer input of every evaluation: the slab is sized for twice the last solve's attempts (at least 48); a solve
        // that needs more ends at that limit and is redone with room for max_attempts
        const int n_limit = keep_tape ? std::min(cap, std::max(48, 2 * h->predicted)) : cap;
        if (keep_tape) { st = ensure_mw_slab(h, 2 + (long long)(h->rk_S - 1) * n_limit, P.Bpad, s); if (st != RNDE_OK) return st; MQ.slab = h->mw_slab; }
        MQ.n_limit = n_limit;
        if (++h->mw_epoch >= 500000u) { h->mw_epoch = 1; HIPCHK(h, hipMemsetAsync(h->mw_xch, 0, (size_t)(cap + 4) * 3 * kMwMeetMax * 8, s)); }
        const int nt = P.Bpad / 16;
        MQ.u_out = u_out_dev; MQ.xch = h->mw_xch; MQ.xcc = h->mw_xcc; MQ.abort_word = h->mw_abort; MQ.epoch = h->mw_epoch; MQ.xch_global = nt > 32 ? 1 : 0; MQ.xcd_slot = h->mw_slot;
        HIPCHK(h, launch_mw<MW_SOLVE>(h, MQ, 0, s));
        if (h->timing) { HIPCHK(h, hipEventRecord(h->tev[1], s)); h->tev_fwd = true; }
        HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->ctl_final, sizeof(StepState), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->h_meta, h->meta, (size_t)cap * sizeof(StepMeta), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->h_init, h->initrec, sizeof(InitRec), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->h_mw_chk, h->mw_abort, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->h_mw_chk + 2, h->mw_xcc, (size_t)nt * 4, hipMemcpyDeviceToHost, s));
        if (h->after_solve) {
            HIPCHK(h, hipEventRecord(h->ev_host, s));
            const rnde_status hs = h->after_solve(s);
            if (hs != RNDE_OK) return hs;
            HIPCHK(h, hipEventSynchronize(h->ev_host));
        } else HIPCHK(h, hipStreamSynchronize(s));
        if (bsweep_failed(h, s)) { h->err = "the one-launch reverse sweep of the previous asynchronous backward call was abandoned: the gradients of that step are invalid (one launch per reversed attempt now in use)"; return RNDE_ERR_HIP; }
        bool bad = h->h_mw_chk[0] != 0;
        for (int i = 1; i < nt && !MQ.xch_global && !bad; ++i) bad = h->h_mw_chk[2 + i] != h->h_mw_chk[2];
        if (bad) {      // a meeting timed out, or the workgroups did not share an XCD: this handle goes back to one launch per attempt, for good
            fprintf(stderr, "[rnde] chain engine: one-launch solve abandoned (%s); one launch per attempted step for the next %d solves\n",
                    h->h_mw_chk[0] ? "a meeting timed out" : "workgroups pinned by block index landed on different XCDs", h->mw_retry_after);
            h->mw_solve = -1; h->mw_clean = 0; ++h->persist_fallbacks;
            hipMemsetAsync(h->mw_abort, 0, 8, s);
            return RNDE_INTERNAL_RETRY;
        }
        if (!h->h_ctl->done && h->h_ctl->n_att >= n_limit && n_limit < cap) { h->predicted = cap; return RNDE_INTERNAL_RETRY; }
        h->pending_bwd = false;
        ++h->one_launch_solves;
        solved = true;
    }
    // ---- stage engine, headline geometry, all workgroups resident at once (<= 32 column tiles): the WHOLE adaptive solve is one launch
    // ---- (rnde_stage_solve.h: weights, uprev and k1 stay in registers across attempts, the error norm meets through agent-scope granules) ----
    if (h->engine == 2 && h->persist == 1 && h->stage_solve && h->sxch && SQ.C <= 32 && n_saveat == 0 && h->n_replay == 0 && !h->couple &&
        SQ.WT == 7 && SQ.HT == 7 && SQ.K2b == 7 && SQ.MT == 49 && SQ.R == 7 && h->D == 784 && h->H == 100 && !h->stage_generic) {
        PersistSync Y{h->tslab, h->pabort, h->pxcc, h->persist_spins};
        HIPCHK(h, slab_prepare(h, SQ.Bpad16, s));
        if (++h->s_epoch >= 500000u) { h->s_epoch = 1; HIPCHK(h, hipMemsetAsync(h->sxch, 0, (size_t)(cap + 1) * 3 * 256 * 8, s)); }
        SolveSync Z{h->sxch, h->s_epoch, cap};
#ifdef RNDE_DIAG
        StageParams SD = SQ;
        if (getenv("RNDE_DIAG_SOLVE")) {      // cycle stamps of workgroup 0, every attempt (tools/diag_solve.py)
            if (h->diag_buf) hipFree(h->diag_buf);
            hipMalloc((void**)&h->diag_buf, 16384);
            hipMemsetAsync(h->diag_buf, 0, 16384, s);
            SD.F.dbg_out = h->diag_buf;
        }
        HIPCHK(h, rnde_launch_stage_solve(&SD, &Y, &Z, h->act2, s));
#else
        HIPCHK(h, rnde_launch_stage_solve(&SQ, &Y, &Z, h->act2, s));
#endif
        if (h->timing) { HIPCHK(h, hipEventRecord(h->tev[1], s)); h->tev_fwd = true; }
        hipLaunchKernelGGL(rnde_stage_finish_kernel, dim3(256), dim3(256), 0, s, SQ, -1, u_out_dev); HIPCHK(h, hipGetLastError());
        const int cnt = std::min(cap, std::max(64, 2 * h->predicted));      // step records copied speculatively; a longer solve fetches the rest below
        HIPCHK(h, hipMemcpyAsync(h->h_mbox, h->mbox, h->mbox_meta_off + (size_t)cnt * sizeof(StepMeta), hipMemcpyDeviceToHost, s));
        if (h->after_solve) {
            HIPCHK(h, hipEventRecord(h->ev_host, s));
            const rnde_status hs = h->after_solve(s);
            if (hs != RNDE_OK) return hs;
            HIPCHK(h, hipEventSynchronize(h->ev_host));
        } else HIPCHK(h, hipStreamSynchronize(s));
#ifdef RNDE_DIAG
        if (getenv("RNDE_DIAG_SOLVE") && h->diag_buf) {
            static unsigned long long hst[1024];
            hipMemcpy(hst, h->diag_buf, 8192, hipMemcpyDeviceToHost);
            int na = 0; while (na < 119 && hst[(na + 1) * 8]) ++na;      // attempts with a successor
            double acc[7] = {0}; int cnt = 0;
            for (int a = 2; a + 1 < na; ++a, ++cnt) {
                const unsigned long long* q = hst + a * 8;
                acc[0] += (double)(q[1] - q[0]); acc[1] += (double)(q[2] - q[1]); acc[2] += (double)(q[3] - q[2]); acc[3] += (double)(q[4] - q[3]);
                acc[4] += (double)(q[5] - q[4]); acc[5] += (double)(q[6] - q[5]); acc[6] += (double)(q[8] - q[0]);
            }
            static unsigned long long arr[512];
            hipMemcpy(arr, (char*)h->diag_buf + 8192, 4096, hipMemcpyDeviceToHost);
            {   // arrival spread and exchange latency of the meeting of attempt 10, all workgroups (100 MHz wall clock: 10 ns units)
                std::vector<long long> a, o; const int nw = SQ.C * SQ.R;
                for (int i = 0; i < nw && i < 256; ++i) if (arr[2 * i]) { a.push_back((long long)arr[2 * i]); o.push_back((long long)arr[2 * i + 1]); }
                if (a.size() > 4) {
                    std::vector<long long> sa = a; std::sort(sa.begin(), sa.end());
                    const long long last = sa.back(), first = sa.front();
                    std::vector<long long> so = o; std::sort(so.begin(), so.end());
                    fprintf(stderr, "meeting of attempt 10, %zu workgroups (x10 ns): arrivals after the first: median %lld, 90%% %lld, last %lld | out after the LAST arrival: first %lld, median %lld, last %lld\n",
                            a.size(), sa[sa.size() / 2] - first, sa[sa.size() * 9 / 10] - first, last - first, so.front() - last, so[so.size() / 2] - last, so.back() - last);
                    int late[7] = {0}; for (size_t i = 0; i < a.size(); ++i) if (a[i] - first > (last - first) * 3 / 4) ++late[(i / SQ.C) % 7];
                    fprintf(stderr, "  latest quarter of the spread by row block: %d %d %d %d %d %d %d\n", late[0], late[1], late[2], late[3], late[4], late[5], late[6]);
                }
            }
            if (cnt) fprintf(stderr, "one-launch solve, workgroup 0, mean over %d attempts (cycles): controller %.0f | START %.0f | stages 1-6 %.0f | reduce + barrier %.0f | "
                                     "meeting %.0f | barrier %.0f | attempt %.0f\n", cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt);
        }
#endif
        if (persist_check_result(h, SQ.C, SQ.R, s)) {
            if (h->pending_bwd) {
                h->pending_bwd = false;
                h->err = "a persistent kernel abandoned its hand-off during or after the previous asynchronous reverse pass: the gradients of that step are invalid (multi-launch kernels now in use)";
                return RNDE_ERR_HIP;
            }
            return RNDE_INTERNAL_RETRY;
        }
        h->pending_bwd = false;
        if (h->h_ctl->n_att > cnt) {
            HIPCHK(h, hipMemcpyAsync(h->h_meta + cnt, h->meta + cnt, (size_t)(h->h_ctl->n_att - cnt) * sizeof(StepMeta), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
        }
        if (!h->h_ctl->done) { h->err = "max_attempts reached"; h->n_att = h->h_ctl->n_att; return RNDE_ERR_MAX_ATTEMPTS; }
        ++h->one_launch_solves;
        solved = true;
    }
    while (!solved) {
        if (h->engine == 3 && h->mw && keep_tape) {   // room in the activation slab for this chunk's evaluations (a regrowth keeps the taped ones)
            st = ensure_mw_slab(h, 2 + (long long)(h->rk_S - 1) * std::min(cap, launched + chunk), P.Bpad, s);
            if (st != RNDE_OK) return st;
            MQ.slab = h->mw_slab;
        }
        for (int i = 0; i < chunk && launched < cap; ++i) {
            if (h->engine == 3 && h->mw) HIPCHK(h, launch_mw<MW_STEP>(h, MQ, launched, s));
            else if (h->engine == 3) HIPCHK(h, launch_chain<CM_STEP>(h, CQ, launched, nullptr, s));
            else if (h->engine == 2) {
#ifdef RNDE_DIAG
                static const int diag_n = getenv("RNDE_DIAG_FWD") ? atoi(getenv("RNDE_DIAG_FWD")) : -1;   // cycle stamps of THIS attempt of a real solve
                if (launched == diag_n) {
                    if (!h->diag_buf) hipMalloc((void**)&h->diag_buf, 8192);
                    hipMemsetAsync(h->diag_buf, 0, 8192, s);
                    StageParams SD = SQ; SD.F.dbg_out = h->diag_buf;
                    HIPCHK(h, stage_attempt(h, SD, launched, s));
                } else
#endif
                HIPCHK(h, stage_attempt(h, SQ, launched, s));
            }
            if ((st = couple_sum(h, P.errpart + (size_t)(launched & 1) * 3 * P.nwg, 3LL * P.nwg, s)) != RNDE_OK) return st;
            ++launched;
        }
        if (h->timing && !h->tev_fwd) { HIPCHK(h, hipEventRecord(h->tev[1], s)); h->tev_fwd = true; }   // (first chunk: normally the whole solve)
        if (h->engine == 3 && h->mw) { MQ.u_out = u_out_dev; HIPCHK(h, launch_mw<MW_FINISH>(h, MQ, launched, s)); }
        else if (h->engine == 3) HIPCHK(h, launch_chain<CM_FINISH>(h, CQ, launched, u_out_dev, s));
        else { hipLaunchKernelGGL(rnde_stage_finish_kernel, dim3(256), dim3(256), 0, s, SQ, launched, u_out_dev); HIPCHK(h, hipGetLastError()); }
        // one synchronisation per chunk: controller state, the persistent kernels' health words, and (speculatively: the solve
        // usually ends in the first chunk) the step metadata and the initial-step record the epilogue needs
        if (h->mbox) {
            HIPCHK(h, hipMemcpyAsync(h->h_mbox, h->mbox, h->mbox_meta_off + (size_t)launched * sizeof(StepMeta), hipMemcpyDeviceToHost, s));
        } else {
            HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->ctl_final, sizeof(StepState), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(h->h_meta, h->meta, (size_t)launched * sizeof(StepMeta), hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipMemcpyAsync(h->h_init, h->initrec, sizeof(InitRec), hipMemcpyDeviceToHost, s));
        }
        if (h->after_solve) {   // (fused training step) wait for the copy only; what the hook queues runs while the host wakes up.  A solve that
            HIPCHK(h, hipEventRecord(h->ev_host, s));   // needs another chunk, or is redone, calls the hook again: it only overwrites its outputs
            const rnde_status hs = h->after_solve(s);
            if (hs != RNDE_OK) return hs;
            HIPCHK(h, hipEventSynchronize(h->ev_host));
        } else HIPCHK(h, hipStreamSynchronize(s));
#ifdef RNDE_DIAG
        if (h->engine == 2 && getenv("RNDE_DIAG_FWD") && h->diag_buf) {
            unsigned long long hst[64] = {0};
            hipMemcpy(hst, h->diag_buf, sizeof(hst), hipMemcpyDeviceToHost);
            if (hst[34]) {
                fprintf(stderr, "attempt %s of a real solve (workgroup 0 thread 0, cycles): controller %lld startC %lld startD %lld |", getenv("RNDE_DIAG_FWD"), (long long)(hst[1]-hst[0]), (long long)(hst[2]-hst[1]), (long long)(hst[3]-hst[2]));
                for (int st_ = 0; st_ < 6; ++st_) fprintf(stderr, " poll %lld", (long long)(hst[4 + 5 * st_] - hst[3 + 5 * st_]));
                fprintf(stderr, " | total %lld  (entry -> all loads issued %lld, -> controller state here %lld)\n", (long long)(hst[34]-hst[0]), (long long)(hst[35]-hst[0]), (long long)(hst[36]-hst[35]));
            }
        }
#endif
        if (h->couple && rnde_comm_health(h->couple) != RNDE_OK) { h->err = std::string("coupled controller: ") + rnde_comm_last_error(h->couple); return RNDE_ERR_HIP; }
        if (bsweep_failed(h, s)) { h->err = "the one-launch reverse sweep of the previous asynchronous backward call was abandoned: the gradients of that step are invalid (one launch per reversed attempt now in use)"; return RNDE_ERR_HIP; }
        if (h->engine == 2 && persist_check_result(h, SQ.C, SQ.R, s)) {
            if (h->pending_bwd) {   // the failure may belong to the asynchronous reverse pass before this forward: its outputs cannot be trusted
                h->pending_bwd = false;
                h->err = "a persistent kernel abandoned its hand-off during or after the previous asynchronous reverse pass: the gradients of that step are invalid (multi-launch kernels now in use)";
                return RNDE_ERR_HIP;
            }
            if (h->couple) {   // a redo on this rank alone would leave the ranks' all-reduce sequences out of step
                h->err = "coupled controller: a persistent kernel abandoned its hand-off; the ranks are out of step -- use cfg.persist = -1 (7-launch kernels) with rnde_node_set_coupling where other work shares the GPU";
                return RNDE_ERR_HIP;
            }
            return RNDE_INTERNAL_RETRY;
        }
        h->pending_bwd = false;
        if (h->h_ctl->done) break;
        if (launched >= cap) { h->err = "max_attempts reached"; h->n_att = h->h_ctl->n_att; return RNDE_ERR_MAX_ATTEMPTS; }
        chunk = h->couple ? 16 : 4;
    }
    h->n_att = h->h_ctl->n_att;
    h->predicted = h->n_att + 1;
    // a hand-off time-out (a co-tenant held CUs for a second, e.g. another process's kernels) must not halve the speed for good:
    // after `persist_retry_after` clean multi-launch solves the one-launch kernels get another chance; each new failure doubles the wait
    if (h->engine == 3 && (h->mw_solve == -1 || h->mw_bsweep == -1) && ++h->mw_clean >= h->mw_retry_after) {      // the same for the chain engine's one-launch kernels
        if (h->mw_solve == -1) h->mw_solve = 1;
        if (h->mw_bsweep == -1) h->mw_bsweep = 1;
        h->mw_retry_after = std::min(1024, 2 * h->mw_retry_after);
        h->mw_slot = (h->mw_slot + 1) & 7;                                           // (and another XCD: the one it was pinned to may be the contended one)
    }
    if (h->persist == -1 && h->engine == 2 && h->cfg.persist >= 0 && ++h->persist_clean >= h->persist_retry_after) {
        h->persist = 1; h->persist_retry_after = std::min(1024, 2 * h->persist_retry_after);
        h->tslab_Bpad = -1;                          // slabs are refilled with the empty pattern before the next persistent launch
    }
    if (nfe_out) *nfe_out = 3 + (int64_t)(h->rk_S - 1) * h->n_att;  // 2 (initial dt) + 1 (fsalfirst) + 6 per attempt (S - 1 for an S-stage table), SURVEY.md B.1-B.2
    // saving callback values (reference neural_ode.jl:116,:126-127): EEst*dt per accepted step
    int nsv = 0;
    h->sv_index.assign(h->n_att, -1);
    // func(u, t, integrator) of the reference: EEst*dt (mnist_node.jl:67), stab*|eigen_est| (:74-79), their blend (:88-97)
    const float stab = 1.0f / 3.5068f;   // 1 / alg_stability_size(Tsit5()), as recalled in SURVEY.md 8a row a9
    auto cbval = [&](float eest, float dt, float eig) -> float {
        const bool eg_ok = !(eig == 0.f || eig != eig);
        switch (h->cfg.regularize) {
            case RNDE_REG_ERR: return eest * dt;
            case RNDE_REG_STIFF: return eg_ok ? stab * fabsf(eig) : 0.f;
            case RNDE_REG_ERR_STIFF: { const float e = eest * dt; return ((e == 0.f || e != e) ? 0.f : e) + 0.1f * (eg_ok ? stab * eig : 0.f); }
            default: return 0.f;
        }
    };
    if (h->cfg.regularize != RNDE_REG_NONE) {
        if (h->cfg.cb_save_start) { if (saveval_host) saveval_host[nsv] = cbval(1.f, 0.f, 1.f); ++nsv; }   // EEst = 1, dt = 0, eigen_est = 1 at init
        for (int i = 0; i < h->n_att; ++i)
            if (h->h_meta[i].flags & F_ACCEPT) {
                if (saveval_host) saveval_host[nsv] = cbval(h->h_meta[i].eest, h->h_meta[i].dt, h->h_meta[i].eigen);
                h->sv_index[i] = nsv++;
            }
    }
    h->n_saveval = nsv;
    if (n_saveval_out) *n_saveval_out = nsv;
    switch (h->h_ctl->status) {
        case 0: break;
        case 2: h->err = "max_attempts reached"; return RNDE_ERR_MAX_ATTEMPTS;
        case 3: h->err = "dt underflow"; return RNDE_ERR_DT_UNDERFLOW;
        default: h->err = "non-finite error estimate or dt"; return RNDE_ERR_NONFINITE;
    }
    h->have_tape = keep_tape != 0;
    return RNDE_OK;
}

extern "C" rnde_status rnde_node_steps(rnde_node* h, float* steps_host, int32_t capacity, int32_t* n_out) {
    if (!h) return RNDE_ERR_BAD_ARG;
    const int n = std::min(capacity, h->n_att);
    for (int i = 0; i < n; ++i) {
        steps_host[4 * i + 0] = h->h_meta[i].t; steps_host[4 * i + 1] = h->h_meta[i].dt;
        steps_host[4 * i + 2] = h->h_meta[i].eest; steps_host[4 * i + 3] = (h->h_meta[i].flags & F_ACCEPT) ? 1.f : 0.f;
    }
    if (n_out) *n_out = h->n_att;
    return RNDE_OK;
}

extern "C" rnde_status rnde_node_set_timing(rnde_node* h, int32_t on) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (on && !h->tev[0]) for (auto& e : h->tev) HIPCHK(h, hipEventCreate(&e));
    h->timing = on ? 1 : 0; h->tev_fwd = h->tev_bwd = false;
    return RNDE_OK;
}
extern "C" rnde_status rnde_node_timing(rnde_node* h, float* fwd_attempts_ms, float* rev_sweep_ms, float* rev_rest_ms) {
    if (!h || !h->timing) return RNDE_ERR_BAD_ARG;
    float a = -1.f, b = -1.f, c = -1.f;
    if (h->tev_fwd) { HIPCHK(h, hipEventSynchronize(h->tev[1])); HIPCHK(h, hipEventElapsedTime(&a, h->tev[0], h->tev[1])); }
    if (h->tev_bwd) {
        HIPCHK(h, hipEventSynchronize(h->tev[4]));
        HIPCHK(h, hipEventElapsedTime(&b, h->tev[2], h->tev[3])); HIPCHK(h, hipEventElapsedTime(&c, h->tev[3], h->tev[4]));
    }
    if (fwd_attempts_ms) *fwd_attempts_ms = a;
    if (rev_sweep_ms) *rev_sweep_ms = b;
    if (rev_rest_ms) *rev_rest_ms = c;
    return RNDE_OK;
}
extern "C" int32_t rnde_node_last_attempts(const rnde_node* h) { return h ? h->n_att : 0; }
extern "C" int32_t rnde_node_fallback_count(const rnde_node* h) { return h ? h->persist_fallbacks : 0; }

extern "C" int32_t rnde_node_one_launch_solves(const rnde_node* h) { return h ? h->one_launch_solves : 0; }
extern "C" int32_t rnde_node_launches_per_attempt(const rnde_node* h) {
    if (!h) return 0;
    return (h->engine == 2 && h->persist != 1) ? 7 : 1;
}

extern "C" rnde_status rnde_node_release_tape(rnde_node* h) {
    if (!h) return RNDE_ERR_BAD_ARG;
    h->have_tape = false;
    return RNDE_OK;
}

extern "C" rnde_status rnde_node_backward_async(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                                float* p_bar_dev, float* tspan_bar_dev, void* stream) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (!h->have_tape) { h->err = "no recorded forward"; return RNDE_ERR_NO_TAPE; }
    if (h->engine == 3) return chain_bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, nullptr, (hipStream_t)stream, false, tspan_bar_dev);
    return bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, nullptr, (hipStream_t)stream, false, tspan_bar_dev);
}

extern "C" rnde_status rnde_node_backward(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host,
                                          float* x_bar_dev, float* p_bar_dev, float* tspan_bar_host, void* stream) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (!h->have_tape) { h->err = "no recorded forward"; return RNDE_ERR_NO_TAPE; }
    if (h->engine == 3) return chain_bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, tspan_bar_host, (hipStream_t)stream);
    return bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, tspan_bar_host, (hipStream_t)stream);
}

// ---- host-pointer variants ------------------------------------------------------------------
extern "C" rnde_status rnde_node_forward_host(rnde_node* h, const float* x, const float* p, int32_t B, float t0, float t1,
                                              float* u_out, int64_t* nfe_out, float* saveval, int32_t* n_saveval_out,
                                              int32_t keep_tape) {
    if (!h) return RNDE_ERR_BAD_ARG;
    float *xd = nullptr, *pd = nullptr, *ud = nullptr;
    const size_t nb = (size_t)h->D * B * 4;
    HIPCHK(h, hipMalloc((void**)&xd, nb)); HIPCHK(h, hipMalloc((void**)&ud, nb)); HIPCHK(h, hipMalloc((void**)&pd, (size_t)h->P * 4));
    HIPCHK(h, hipMemcpy(xd, x, nb, hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(pd, p, (size_t)h->P * 4, hipMemcpyHostToDevice));
    rnde_status st = rnde_node_forward(h, xd, pd, B, t0, t1, ud, nfe_out, saveval, n_saveval_out, keep_tape, nullptr);
    if (st == RNDE_OK || st == RNDE_ERR_MAX_ATTEMPTS) hipMemcpy(u_out, ud, nb, hipMemcpyDeviceToHost);
    hipFree(xd); hipFree(pd); hipFree(ud);
    return st;
}
extern "C" rnde_status rnde_node_backward_host(rnde_node* h, const float* u_bar, const float* saveval_bar, float* x_bar,
                                               float* p_bar, float* tspan_bar) {
    if (!h) return RNDE_ERR_BAD_ARG;
    float *ub = nullptr, *xb = nullptr, *pb = nullptr;
    const size_t nb = (size_t)h->D * h->B * 4;
    HIPCHK(h, hipMalloc((void**)&ub, nb)); HIPCHK(h, hipMalloc((void**)&xb, nb)); HIPCHK(h, hipMalloc((void**)&pb, (size_t)h->P * 4));
    HIPCHK(h, hipMemcpy(ub, u_bar, nb, hipMemcpyHostToDevice));
    rnde_status st = rnde_node_backward(h, ub, saveval_bar, xb, pb, tspan_bar, nullptr);
    if (st == RNDE_OK) { hipMemcpy(x_bar, xb, nb, hipMemcpyDeviceToHost); hipMemcpy(p_bar, pb, (size_t)h->P * 4, hipMemcpyDeviceToHost); }
    hipFree(ub); hipFree(xb); hipFree(pb);
    return st;
}

// ---- kernel-level entry points ----------------------------------------------------------------
extern "C" rnde_status rnde_debug_feval(rnde_node* h, const float* u_dev, const float* p_dev, int32_t B, float t,
                                        float* out_dev, void* stream) {
    if (!h || B < 1 || B > h->cfg.max_batch) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    StepParams P = make_params(h, u_dev, B, 0.f, 1.f, 0);
    P.forced = 1; P.forced_t = t; P.dbg_out = out_dev;
    if (h->engine == 3) {
        rnde_status st3 = chain_pack(h, p_dev, s);
        if (st3 != RNDE_OK) return st3;
        if (h->mw) HIPCHK(h, launch_mw<MW_FEVAL>(h, make_mw_params(h, P), 0, s));
        else HIPCHK(h, launch_chain<CM_FEVAL>(h, make_chain_params(h, P), 0, nullptr, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return RNDE_OK;
    }
    if (h->engine == 2) {
        rnde_status st2 = stage_pack_weights(h, p_dev, s);
        if (st2 != RNDE_OK) return st2;
        StageParams SQ = make_stage_params(h, P, p_dev);
        HIPCHK(h, launch_stage<SM_FEVAL1>(h, SQ, 0, 0, s));
        HIPCHK(h, launch_stage<SM_FEVAL2>(h, SQ, 0, 0, s));
        HIPCHK(h, hipStreamSynchronize(s));
        return RNDE_OK;
    }
    h->err = "rnde_debug_feval: unknown engine";
    return RNDE_ERR_BAD_ARG;
}

extern "C" rnde_status rnde_debug_attempt(rnde_node* h, const float* uprev_dev, const float* k1_dev, const float* p_dev,
                                          int32_t B, float t, float dt, float* k_out_dev, float* unew_out_dev,
                                          float* eest_out, void* stream) {
    if (!h || B < 1 || B > h->cfg.max_batch) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    h->have_tape = false;
    StepParams P = make_params(h, uprev_dev, B, 0.f, 1.f, 0);
    P.forced = 1; P.forced_t = t; P.forced_dt = dt;
    if (h->engine == 3) {
        rnde_status st3 = chain_pack(h, p_dev, s);
        if (st3 != RNDE_OK) return st3;
        const ChainParams CQ = make_chain_params(h, P);
        const int nks = h->NKD;
        HIPCHK(h, chain_convert(k1_dev, h->f0, h->D, B, CQ.ntiles, nks, 0, s));
        if (h->mw) { HIPCHK(h, launch_mw<MW_STEP>(h, make_mw_params(h, P), 0, s)); HIPCHK(h, launch_mw<MW_FINISH>(h, make_mw_params(h, P), 1, s)); }
        else { HIPCHK(h, launch_chain<CM_STEP>(h, CQ, 0, nullptr, s)); HIPCHK(h, launch_chain<CM_FINISH>(h, CQ, 1, nullptr, s)); }
        HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->ctl_final, sizeof(StepState), hipMemcpyDeviceToHost, s));
        const ChainRec CL{(long long)CQ.ntiles * nks * 64, h->rk_S};
        for (int sidx = 2; sidx <= h->rk_S; ++sidx)      // (k_out: rk_S - 1 arrays)
            HIPCHK(h, chain_convert(h->arena + CL.k(sidx), k_out_dev + (size_t)(sidx - 2) * h->D * B, h->D, B, CQ.ntiles, nks, 1, s));
        HIPCHK(h, chain_convert(h->arena + CL.unew(), unew_out_dev, h->D, B, CQ.ntiles, nks, 1, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (eest_out) *eest_out = h->h_ctl->last_eest;
        return RNDE_OK;
    }
    // k1 goes to the f0 buffer (column stride D in both layouts)
    HIPCHK(h, hipMemsetAsync(h->f0, 0, (size_t)h->D * P.Bpad * 4, s));
    HIPCHK(h, hipMemcpyAsync(h->f0, k1_dev, (size_t)h->D * B * 4, hipMemcpyDeviceToDevice, s));
    {
        rnde_status st2 = stage_pack_weights(h, p_dev, s);
        if (st2 != RNDE_OK) return st2;
        StageParams SQ = make_stage_params(h, P, p_dev);
        HIPCHK(h, stage_attempt(h, SQ, 0, s));
        hipLaunchKernelGGL(rnde_stage_finish_kernel, dim3(256), dim3(256), 0, s, SQ, 1, (float*)nullptr);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemcpyAsync(h->h_ctl, h->ctl_final, sizeof(StepState), hipMemcpyDeviceToHost, s));
    RecLayout L{(long long)h->D * P.Bpad, (long long)h->H * P.Bpad};
    const float* R = h->arena;  // record 0 (no-tape: live == -1 -> rec 0)
    for (int sidx = 2; sidx <= 7; ++sidx)
        HIPCHK(h, hipMemcpyAsync(k_out_dev + (size_t)(sidx - 2) * h->D * B, R + L.k(sidx), (size_t)h->D * B * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemcpyAsync(unew_out_dev, R + L.unew(), (size_t)h->D * B * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (h->engine == 2 && persist_failed(h, h->sR * (P.Bpad / 16), P.Bpad / 16, h->sR, s)) { h->err = "persistent attempt kernel abandoned its hand-off; call again (multi-launch kernels now in use)"; return RNDE_ERR_HIP; }
    if (eest_out) *eest_out = h->h_ctl->last_eest;
    return RNDE_OK;
}

static rnde_status bench_attempt_impl(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters, int32_t taped,
                                      float* mean_us_out, void* stream);
extern "C" rnde_status rnde_bench_attempt(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters,
                                          float* mean_us_out, void* stream) {
    return bench_attempt_impl(h, x_dev, p_dev, B, iters, 0, mean_us_out, stream);
}
extern "C" rnde_status rnde_bench_attempt_taped(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters,
                                                float* mean_us_out, void* stream) {
    return bench_attempt_impl(h, x_dev, p_dev, B, iters, 1, mean_us_out, stream);
}
extern "C" rnde_status rnde_bench_attempt_cold_tape(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters,
                                                    int32_t records, float* mean_us_out, void* stream) {
    if (records < 2) return RNDE_ERR_BAD_ARG;
    return bench_attempt_impl(h, x_dev, p_dev, B, iters, records, mean_us_out, stream);
}
static rnde_status bench_attempt_impl(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, int32_t iters, int32_t taped,
                                      float* mean_us_out, void* stream) {
    if (!h || B < 1 || B > h->cfg.max_batch || iters < 1) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    h->have_tape = false;
    if (taped > 1) { const rnde_status sa = ensure_arena(h, std::min<long long>(taped, h->cfg.max_attempts)); if (sa != RNDE_OK) return sa; }
    StepParams P = make_params(h, x_dev, B, 0.f, 1.f, taped ? 1 : 0);   // taped: the variant a training step runs (record 0 of the arena)
    P.forced = 1; P.forced_t = 0.f; P.forced_dt = 0.05f;
    rnde_status st = h->engine == 3 ? chain_pack(h, p_dev, s) : RNDE_OK;
    if (st != RNDE_OK) return st;
    StageParams SQ{};
    ChainParams CQ{};
    MwParams MQ{};
    if (h->engine == 3 && h->mw) {
        if (taped) { st = ensure_mw_slab(h, 8, P.Bpad, s); if (st != RNDE_OK) return st; }
        MQ = make_mw_params(h, P);
        HIPCHK(h, launch_mw<MW_INIT_A>(h, MQ, 0, s));   // k1 = f(x, 0) into f0
        for (int i = 0; i < 3; ++i) HIPCHK(h, launch_mw<MW_STEP>(h, MQ, 0, s));
    } else if (h->engine == 3) {
        CQ = make_chain_params(h, P);
        HIPCHK(h, launch_chain<CM_INIT_A>(h, CQ, 0, nullptr, s));   // k1 = f(x, 0) into f0
        for (int i = 0; i < 3; ++i) HIPCHK(h, launch_chain<CM_STEP>(h, CQ, 0, nullptr, s));
    } else {
        st = stage_pack_weights(h, p_dev, s);
        if (st != RNDE_OK) return st;
        SQ = make_stage_params(h, P, p_dev);
        HIPCHK(h, launch_stage<SM_I1>(h, SQ, 0, 0, s));
        HIPCHK(h, launch_stage<SM_I2>(h, SQ, 0, 0, s));   // k1 = f(x, 0) into f0
        for (int i = 0; i < 3; ++i) HIPCHK(h, stage_attempt(h, SQ, 0, s));
    }
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, s));
    const auto host_t0 = std::chrono::steady_clock::now();
    // taped > 1 (stage engine): every attempt writes ANOTHER record of the arena, `taped` of them in turn -- what a solve does (31 records
    // of 21.7 MB at B = 512: the tape leaves the chip); taped == 1 rewrites record 0, which then lives in the Infinity Cache
    const int cyc = (taped > 1 && h->engine == 2) ? (int)std::min<long long>(taped, h->arena_recs) : 1;
    for (int i = 0; i < iters; ++i) {
        if (h->engine == 3 && h->mw) HIPCHK(h, launch_mw<MW_STEP>(h, MQ, 0, s));
        else if (h->engine == 3) HIPCHK(h, launch_chain<CM_STEP>(h, CQ, 0, nullptr, s));
        else { SQ.F.rec_shift = i % cyc; HIPCHK(h, stage_attempt(h, SQ, 0, s)); }
    }
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count();
    if (getenv("RNDE_TRACE_HOST")) fprintf(stderr, "[rnde] host enqueue: %.2f us per attempt (%d launches each)\n", host_us / iters, h->engine == 2 ? 7 : 1);
    HIPCHK(h, hipEventRecord(e1, s));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (mean_us_out) *mean_us_out = ms * 1000.f / iters;
#ifdef RNDE_DIAG
    {   // phase stamps of the LAST f evaluation of one launch (workgroup 0), in shader cycles relative to stamp 0 of wave 0
        unsigned long long* d = nullptr; unsigned long long hst[512] = {0};
        hipMalloc((void**)&d, sizeof(hst)); hipMemset(d, 0, sizeof(hst));
        P.dbg_out = (float*)d;
        if (h->engine == 3 && h->mw) {
            MQ.F.dbg_out = (float*)d; launch_mw<MW_STEP>(h, MQ, 0, s); hipStreamSynchronize(s); hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
            fprintf(stderr, "mw chain stamps (cycles): LDS fill %lld, controller %lld, state load %lld |", (long long)(hst[1]-hst[0]), (long long)(hst[2]-hst[1]), (long long)(hst[3]-hst[2]));
            for (int i = 4; i <= 9; ++i) fprintf(stderr, " eval%d %lld", i - 3, (long long)(hst[i]-hst[i-1]));
            fprintf(stderr, " | tail %lld total %lld\n  first eval: input->L0 %lld", (long long)(hst[10]-hst[9]), (long long)(hst[10]-hst[0]), (long long)(hst[16]-hst[3]));
            for (int l = 0; l < h->mg.n_layers; ++l) fprintf(stderr, " L%d %lld", l, (long long)(hst[17+l]-hst[16+l]));
            fprintf(stderr, "\n");
            hipFree(d); return RNDE_OK; }
        if (h->engine == 3) { CQ.F.dbg_out = (float*)d; launch_chain<CM_STEP>(h, CQ, 0, nullptr, s); hipStreamSynchronize(s); hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
            fprintf(stderr, "chain stamps (cycles): fill %lld ctl %lld loads %lld |", (long long)(hst[1]-hst[0]), (long long)(hst[2]-hst[1]), (long long)(hst[3]-hst[2]));
            for (int i = 4; i <= 9; ++i) fprintf(stderr, " st%d %lld", i - 3, (long long)(hst[i]-hst[i-1]));
            fprintf(stderr, " | tail %lld total %lld\n", (long long)(hst[10]-hst[9]), (long long)(hst[10]-hst[0]));
            for (int l = 0; l < h->cg.n_layers; ++l) fprintf(stderr, "  layer %d: bias %lld mm %lld act %lld (gap to next %lld)\n", l, (long long)(hst[17+4*l]-hst[16+4*l]), (long long)(hst[18+4*l]-hst[17+4*l]), (long long)(hst[19+4*l]-hst[18+4*l]), l + 1 < h->cg.n_layers ? (long long)(hst[20+4*l]-hst[19+4*l]) : 0LL);
            hipFree(d); return RNDE_OK; }
        if (h->engine == 2 && h->persist == 1) {
            SQ.F.dbg_out = (float*)d; stage_attempt(h, SQ, 0, s); hipStreamSynchronize(s);
            hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost); hipFree(d);
            fprintf(stderr, "persistent attempt (workgroup 0 thread 0, cycles): prologue(weights) %lld ctl %lld startC %lld startD %lld\n", 0LL, (long long)(hst[1]-hst[0]), (long long)(hst[2]-hst[1]), (long long)(hst[3]-hst[2]));
            for (int st = 1; st <= 6; ++st) { const unsigned long long* q = hst + 4 + 5 * (st - 1); const unsigned long long prev = st == 1 ? hst[3] : hst[8 + 5 * (st - 2)];
                if (st < 6) fprintf(stderr, "  stage %d: poll %lld A %lld B %lld C %lld D+put %lld\n", st, (long long)(q[0]-prev), (long long)(q[1]-q[0]), (long long)(q[2]-q[1]), (long long)(q[3]-q[2]), (long long)(q[4]-q[3]));
                else fprintf(stderr, "  stage %d: poll %lld A %lld B %lld C+err %lld | total %lld cycles\n", st, (long long)(q[0]-prev), (long long)(q[1]-q[0]), (long long)(q[2]-q[1]), (long long)(hst[34]-q[2]), (long long)(hst[34]-hst[0])); }
            fprintf(stderr, "per wave, relative to wave 0's poll-done of the stage: poll-done | barrier A in, out | B done | C done (before the barrier of D) | after it | put done\n");
            for (int st = 1; st <= 5; ++st) for (int w = 0; w < 7; ++w) { const unsigned long long* q = hst + 64 + ((st - 1) * 8 + w) * 8; const long long z = (long long)hst[64 + (st - 1) * 64];
                fprintf(stderr, "  stage %d wave %d: %6lld | %6lld %6lld | %6lld | %6lld | %6lld | %6lld\n", st, w, (long long)q[0]-z, (long long)q[1]-z, (long long)q[2]-z, (long long)q[3]-z, (long long)q[4]-z, (long long)q[5]-z, (long long)q[6]-z); }
            return RNDE_OK;
        }
        SQ.F.dbg_out = (float*)d; stage_attempt(h, SQ, 0, s);
        hipStreamSynchronize(s);
        hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost); hipFree(d);
        fprintf(stderr, "stamps (cycles since wave0 stamp0); column-owner: start sync1 gemm1 sync2 reduce sync3 gemm2 tanh | stage: start scalars A sync B C sync D\n");
        for (int w = 0; w < 8; ++w) { fprintf(stderr, "wave %d:", w); for (int i = 0; i < 8; ++i) fprintf(stderr, " %7lld", (long long)(hst[w * 8 + i] - hst[0])); fprintf(stderr, "\n"); }
    }
#endif
    return RNDE_OK;
}

// ---- reverse pass driver ------------------------------------------------------------------------
static rnde_status bwd_prepare(rnde_node* h) {
    BwdBuffers& b = h->bw;
    if (b.ready) return RNDE_OK;
    const size_t A = (size_t)h->D * h->Bpad_max, HB = (size_t)h->H * h->Bpad_max;
    const int cap = h->cfg.max_attempts;
    HIPCHK(h, hipMalloc((void**)&b.U, A * 4)); HIPCHK(h, hipMalloc((void**)&b.K1, A * 4)); HIPCHK(h, hipMalloc((void**)&b.UB1, A * 4));
    HIPCHK(h, hipMalloc((void**)&b.zi2, 2 * A * 4)); HIPCHK(h, hipMalloc((void**)&b.zi1, 2 * HB * 4));
    HIPCHK(h, hipMalloc((void**)&b.svb_att, (size_t)cap * 4));
    HIPCHK(h, hipMalloc((void**)&b.bstate, 2 * sizeof(BState))); HIPCHK(h, hipMalloc((void**)&b.ibstate, 2 * sizeof(IBState)));
    HIPCHK(h, hipMalloc((void**)&b.bpart, (size_t)(2 * h->nwg_max + 256) * 4 * 4)); HIPCHK(h, hipMalloc((void**)&b.ipart, (size_t)2 * h->nwg_max * 4 * 4));   // (+256 entries: finish_attempt_scalars reads whole 256-entry blocks)
    HIPCHK(h, hipMalloc((void**)&b.tspan_out, 2 * 4));
    const size_t nev = (size_t)6 * cap + 2;
    // (ev1 / h_ev1 are sized for [ev1 | ev2 | svb] back to back: bwd_run lays the three out contiguously and sends them in ONE copy)
    const size_t desc_blob = 2 * nev * sizeof(EvalDesc) + (size_t)cap * 4 + 64;
    HIPCHK(h, hipMalloc((void**)&b.ev1, desc_blob)); HIPCHK(h, hipMalloc((void**)&b.ev2, nev * sizeof(EvalDesc)));
    HIPCHK(h, hipHostMalloc((void**)&b.h_ev1, desc_blob)); HIPCHK(h, hipHostMalloc((void**)&b.h_ev2, nev * sizeof(EvalDesc)));
    HIPCHK(h, hipHostMalloc((void**)&b.h_svb, (size_t)cap * 4));
    const size_t seg = std::max((size_t)h->H * (h->D + 2), (size_t)h->D * (h->H + 2));
    b.slab_floats = seg * 256;              // per layer; two layers back to back
    HIPCHK(h, hipMalloc((void**)&b.slab, 2 * b.slab_floats * 4));
    HIPCHK(h, hipMalloc((void**)&b.slab_r, 2 * 16 * seg * 4));      // second-level partials, one region per layer
    if (!h->wstream) {
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        HIPCHK(h, hipStreamCreateWithPriority(&h->wstream, hipStreamNonBlocking, prio_least));   // never ahead of the sweep
        h->wevents.resize(66);
        for (auto& e : h->wevents) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (h->engine == 2) {
        HIPCHK(h, hipMalloc((void**)&b.UTB, A * 4)); HIPCHK(h, hipMalloc((void**)&b.UNB, A * 4)); HIPCHK(h, hipMalloc((void**)&b.UPB0, A * 4));
        HIPCHK(h, hipMalloc((void**)&b.GB, 15 * A * 4));   // gbar_1..6, EXK, EXG (stiffness extras), W_1..7 (saveat)
    }
    b.ready = true;
    return RNDE_OK;
}

static bool wgrad3_ok(int M, int Nx) {
    const bool tall = M >= Nx;
    const int wide = tall ? M : Nx + 2, narrow = tall ? Nx + 2 : M;
    return getenv("RNDE_WGRAD_LEGACY") == nullptr && getenv("RNDE_WGRAD_V2") == nullptr && M % 4 == 0 && Nx % 4 == 0 && wide > 656 &&
           wide <= 800 && narrow <= 112;
}
// `max_chunks` > 0 caps the number of chunks (two workgroups each) of the 16x16x4 kernel: 16 for the launches that run
// underneath the sweep on the CUs it leaves idle, see bwd_run
static rnde_status launch_wgrad_part(rnde_node* h, const EvalDesc* ev, int n_evals, int per_chunk, int M, int Nx, int Bpad,
                                     float* slab, int* chunk_cursor, hipStream_t s, int max_chunks = 0) {
    if (n_evals <= 0) return RNDE_OK;
    const int mtiles = (M + 31) / 32, ntiles = (Nx + 2 + 31) / 32;
    const bool tall = M >= Nx;  // layer 2: M = D; layer 1: M = H
    const int MB = tall ? 2 : 4, NB = tall ? 4 : 2;
    const int blocks = ((mtiles + MB - 1) / MB) * ((ntiles + NB - 1) / NB);
    const long long len = (long long)M * (Nx + 2);
    const int chunks = (n_evals + per_chunk - 1) / per_chunk;      // (chunking of the direct-from-global kernels only; checked where they are launched)
    float* dst = slab + (size_t)(*chunk_cursor) * len;
    static const bool legacy = getenv("RNDE_WGRAD_LEGACY") != nullptr;   // direct-from-global variant, kept for A/B runs
    const bool fits = tall ? (Nx + 2 <= 128) : (M <= 128);                // the staged kernel covers 128 on the un-split side
    if (wgrad3_ok(M, Nx)) {                                               // (RNDE_WGRAD_V2, read per call, keeps the 32x32x2 staged kernel: A/B)
        // 16x16x4 kernel: two workgroups (the halves of the wide side) per chunk of 32-column steps
        const int total_steps = n_evals * ((Bpad + 31) / 32);
        static const int target_chunks = getenv("RNDE_WGRAD3_CHUNKS") ? atoi(getenv("RNDE_WGRAD3_CHUNKS")) : 128;
        int sc = std::max(1, std::min({max_chunks > 0 ? max_chunks : target_chunks, total_steps, 256}));
        const int steps_per_chunk = (total_steps + sc - 1) / sc;
        sc = (total_steps + steps_per_chunk - 1) / steps_per_chunk;
        if ((size_t)(*chunk_cursor + sc) * (size_t)len > h->bw.slab_floats) { h->err = "weight-gradient slab overflow"; return RNDE_ERR_BAD_ARG; }
        const size_t lds = (size_t)2 * 32 * (464 + 144) * sizeof(float);   // two buffers
        static const hipError_t attr = [&] {
            hipError_t e = hipFuncSetAttribute((const void*)rnde_wgrad3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return e == hipSuccess ? hipFuncSetAttribute((const void*)rnde_wgrad3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : e;
        }();
        HIPCHK(h, attr);
        if (tall) hipLaunchKernelGGL((rnde_wgrad3_kernel<true>), dim3(2, sc), dim3(448), lds, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
        else hipLaunchKernelGGL((rnde_wgrad3_kernel<false>), dim3(2, sc), dim3(448), lds, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
        HIPCHK(h, hipGetLastError());
        *chunk_cursor += sc;
        return RNDE_OK;
    }
    if (!legacy && fits) {
        // staged kernel: chunks are ranges of 32-column steps; pick the count that fills the chip in whole rounds
        // (3 workgroups per CU -> 768 resident: one round; measured 768 / 1100 / 1536 / 1792 -> 4.44 / 4.57 / 4.51 / 4.54 ms per step)
        const int pblocks = tall ? (M + 127) / 128 : (Nx + 2 + 127) / 128;
        const int total_steps = n_evals * ((Bpad + 31) / 32);
        static const int target_wgs = getenv("RNDE_WGRAD_WGS") ? atoi(getenv("RNDE_WGRAD_WGS")) : 768;
        int sc = std::max(1, std::min({target_wgs / pblocks, total_steps, 256}));
        const int steps_per_chunk = (total_steps + sc - 1) / sc;
        sc = (total_steps + steps_per_chunk - 1) / steps_per_chunk;
        if ((size_t)(*chunk_cursor + sc) * (size_t)len > h->bw.slab_floats) { h->err = "weight-gradient slab overflow"; return RNDE_ERR_BAD_ARG; }
        if (tall) hipLaunchKernelGGL((rnde_wgrad2_kernel<true>), dim3(pblocks, sc), dim3(256), 0, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
        else hipLaunchKernelGGL((rnde_wgrad2_kernel<false>), dim3(pblocks, sc), dim3(256), 0, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
        HIPCHK(h, hipGetLastError());
        *chunk_cursor += sc;
        return RNDE_OK;
    } else {
        // (this check used to sit in front of ALL paths with the direct kernels' chunk count -- up to 240 -- and refused solves of more than ~65
        //  attempts on the 16x16x4 path, which needs 127 chunks behind the sweep: "slab overflow" in a training run whose step count had grown)
        if ((size_t)(*chunk_cursor + chunks) * (size_t)len > h->bw.slab_floats) { h->err = "weight-gradient slab overflow"; return RNDE_ERR_BAD_ARG; }
        if (tall) hipLaunchKernelGGL((rnde_wgrad_kernel<2, 4>), dim3(blocks, chunks), dim3(64), 0, s, ev, n_evals, per_chunk, M, Nx, Bpad, dst);
        else hipLaunchKernelGGL((rnde_wgrad_kernel<4, 2>), dim3(blocks, chunks), dim3(64), 0, s, ev, n_evals, per_chunk, M, Nx, Bpad, dst);
    }
    HIPCHK(h, hipGetLastError());
    *chunk_cursor += chunks;
    return RNDE_OK;
}
static rnde_status launch_wgrad_reduce(rnde_node* h, const float* slab, int chunks, int M, int Nx, float* out, hipStream_t s) {
    const long long len = (long long)M * (Nx + 2);
    const int grid = (int)std::min<long long>((len + 255) / 256, 2048);
    if (chunks <= 16) {
        hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, slab, chunks, chunks, len, out);
    } else {   // two passes: 16 chunk groups in parallel, then their 16 partial sums (fixed order => deterministic)
        const int per_group = (chunks + 15) / 16, groups = (chunks + per_group - 1) / per_group;
        hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, groups), dim3(256), 0, s, slab, chunks, per_group, len, h->bw.slab_r);
        hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)h->bw.slab_r, groups, groups, len, out);
    }
    HIPCHK(h, hipGetLastError());
    return RNDE_OK;
}

static rnde_status bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                           float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync, float* tspan_bar_dev) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    rnde_status st = bwd_prepare(h);
    if (st != RNDE_OK) return st;
    BwdBuffers& b = h->bw;
    int n_att = h->n_att;
    // (coupled controller: every rank passes the cotangent of its own loss; the shared scalars then carry `world` times the
    //  single-device cotangent, like everything else -- see rnde_node_set_coupling)
    const float svb_scale = h->couple ? (float)h->couple_world : 1.f;
    for (int i = 0; i < n_att; ++i)
        b.h_svb[i] = (saveval_bar_host && h->sv_index[i] >= 0) ? svb_scale * saveval_bar_host[h->sv_index[i]] : 0.f;
    // one host-to-device copy for everything the reverse pass reads from the host: [ev1 (ne) | ev2 (ne) | svb (n_att)], ne = 2 + 6 n_att
    const int ne_all = 2 + 6 * n_att;
    EvalDesc* const h_ev1 = b.h_ev1; EvalDesc* const h_ev2 = b.h_ev1 + ne_all; float* const h_svb_blob = (float*)(b.h_ev1 + 2 * (size_t)ne_all);
    EvalDesc* const d_ev1 = b.ev1;   EvalDesc* const d_ev2 = b.ev1 + ne_all;   float* const d_svb = (float*)(b.ev1 + 2 * (size_t)ne_all);
    memcpy(h_svb_blob, b.h_svb, (size_t)n_att * 4);
    BwdParams Q{};
    Q.F = make_params(h, h->xcopy, h->B, h->t0, h->t1, 1);
    Q.U = b.U; Q.K1 = b.K1; Q.UB1 = b.UB1; Q.zi2 = b.zi2; Q.zi1 = b.zi1; Q.svb_att = d_svb;
    Q.bstate = b.bstate; Q.ibstate = b.ibstate; Q.bpart = b.bpart; Q.ipart = b.ipart;
    Q.ubar = u_bar_dev; Q.xbar = x_bar_dev; Q.tspan_out = b.tspan_out;
    Q.n_att = n_att; Q.track_ctrl = h->cfg.track_ctrl; Q.track_initdt = h->cfg.track_initdt; Q.reg_kind = h->cfg.regularize;
    Q.bpart_n = Q.F.nwg;
    Q.tspan_scale = h->couple ? 1.f / (float)h->couple_world : 1.f;
#ifdef RNDE_DIAG
    if (getenv("RNDE_DIAG_BWD")) { if (!h->diag_buf) hipMalloc((void**)&h->diag_buf, 8192); hipMemset(h->diag_buf, 0, 512); Q.F.dbg_out = h->diag_buf; }
#endif
    Q.sv_T = (int)h->saveat.size();
    Q.sv_ubar0 = (!h->saveat.empty() && h->saveat[0] == h->t0) ? u_bar_dev : nullptr;
    if (!h->saveat.empty() && h->engine != 2) { h->err = "saveat reverse pass runs on the stage engine only"; return RNDE_ERR_BAD_ARG; }
    // ---- evaluation descriptors for the parameter-gradient GEMMs (all pointers are known before the sweep) ----
    const long long A = (long long)h->D * Q.F.Bpad, HB = (long long)h->H * Q.F.Bpad;
    RecLayout L{A, HB};
    // order: the two evaluations of the initial-step heuristic, then 6 per attempt -- so that "everything up to attempt n" is
    // one contiguous range for the launch that runs after the sweep
    int ne = 2;
    h_ev2[0] = EvalDesc{b.zi2, h->h0, h->t0, 0};           h_ev1[0] = EvalDesc{b.zi1, h->xcopy, h->t0, 0};
    h_ev2[1] = EvalDesc{b.zi2 + A, h->h1, h->t0 + h->h_init->dt0, 0}; h_ev1[1] = EvalDesc{b.zi1 + HB, h->u1, h->t0 + h->h_init->dt0, 0};
    for (int n = 0; n < n_att; ++n) {
        const StepMeta& m = h->h_meta[n];
        const float* R = h->arena + (long long)m.rec * h->rec_stride;
        for (int sidx = 2; sidx <= 7; ++sidx) {
            const float ts = m.t + tsC(sidx - 1) * m.dt;
            h_ev2[ne] = EvalDesc{R + L.k(sidx), R + L.h(sidx), ts, 0};
            h_ev1[ne] = EvalDesc{R + L.z1(sidx), sidx < 7 ? R + L.g(sidx) : R + L.unew(), ts, 0};
            ++ne;
        }
    }
    HIPCHK(h, hipMemcpyAsync(b.ev1, b.h_ev1, 2 * (size_t)ne * sizeof(EvalDesc) + (size_t)n_att * 4, hipMemcpyHostToDevice, s));
    float* slab1 = b.slab;
    float* slab2w = b.slab + b.slab_floats;
    int cur1 = 0, cur2 = 0, evi = 0;
    const int per_chunk = std::max(1, (ne + 239) / 240);          // (chunking of the 32x32x2 kernels; the 16x16x4 kernel chunks by steps)
    // ---- weight-gradient GEMMs underneath the sweep ----
    // The persistent reverse kernel occupies 8 * R * ceil(C / 8) CUs (224 of 256 at B = 512: one workgroup per CU, 28 per XCD)
    // and is latency bound; the 32 CUs it cannot use sit idle for the whole sweep (~1.7 ms).  A launch of 16 chunks x 2
    // workgroups of rnde_wgrad3_kernel lands 4 per XCD (round-robin dispatch) and, at 155 KB of LDS and 238 VGPRs per workgroup,
    // exactly one per CU -- it takes those idle CUs and nothing else.  So the evaluations of the attempts reversed first
    // (`side_frac` of them, in groups) go to a second stream as such 32-workgroup launches, each waiting on an event recorded
    // after its last reverse launch; the rest runs on all CUs after the sweep as before.  An earlier form of this overlap with
    // unrestricted grids was a net loss (the GEMM waves took CUs the sweep's workgroups needed: 6.4 -> 8.4..9.8 ms per step).
    const int side_pct = h->wgrad_side_pct;
    const int sweep_cus = 8 * h->sR * ((Q.F.Bpad / 16 + 7) / 8);
    const bool side = h->engine == 2 && h->persist == 1 && side_pct > 0 && n_att >= 8 && sweep_cus <= 224 &&
                      wgrad3_ok(h->H, h->D) && wgrad3_ok(h->D, h->H);
    const int side_att = side ? std::min(n_att, n_att * side_pct / 100) : 0;      // attempts [n_att - side_att, n_att)
    const int group = std::max(4, (side_att + 5) / 6);                             // <= 6 side launches per layer (slab space: 16 chunks each)
    bool used_side = false;
    auto wgrad_group = [&](int lo, int hi, bool on_side) -> rnde_status {
        if (hi <= lo) return RNDE_OK;
        hipStream_t ws = s;
        if (on_side) {
            hipEvent_t ev = h->wevents[evi++ % 64];
            HIPCHK(h, hipEventRecord(ev, s));
            HIPCHK(h, hipStreamWaitEvent(h->wstream, ev, 0));
            ws = h->wstream; used_side = true;
        }
        rnde_status r = launch_wgrad_part(h, d_ev1 + lo, hi - lo, per_chunk, h->H, h->D, h->B, slab1, &cur1, ws, on_side ? 16 : 0);
        if (r != RNDE_OK) return r;
        return launch_wgrad_part(h, d_ev2 + lo, hi - lo, per_chunk, h->D, h->H, h->B, slab2w, &cur2, ws, on_side ? 16 : 0);
    };
    int hi_att = n_att;                                           // evaluations of attempts >= hi_att are already launched
    hipError_t e;
    h->tev_bwd = false;
    if (h->timing) HIPCHK(h, hipEventRecord(h->tev[2], s));
    if (h->engine == 2) {
        // stage engine sweep: one persistent launch per reversed attempt (fallback: 7 launches); then the (column-owner) kernels for the initialisation part
        if (!h->rev_packed) {
            HIPCHK(h, stage_pack(h, h->pcopy, h->spwBt, 2, h->sMT, h->sKHb, s));
            HIPCHK(h, stage_pack(h, h->pcopy, h->spwDt, 3, h->sHT, h->sMT, s));
        }
        h->rev_packed = false;
        BStageParams BQ{};
        BQ.B = Q; BQ.p = h->pcopy; BQ.pwBt = h->spwBt; BQ.pwDt = h->spwDt; BQ.slab = h->slab2;
        BQ.UTB = b.UTB; BQ.UNB = b.UNB; BQ.UPB0 = b.UPB0; BQ.GB = b.GB;
        BQ.EXK = b.GB + 6 * A; BQ.EXG = b.GB + 7 * A; BQ.SVW = b.GB + 8 * A;
        BQ.sv_t = h->saveat.empty() ? nullptr : h->sv_t_dev; BQ.sv_ubar = u_bar_dev; BQ.nsave = (int)h->saveat.size();
        BQ.MT = h->sMT; BQ.WT = h->sWT; BQ.R = h->sR; BQ.C = Q.F.Bpad / 16; BQ.HT = h->sHT; BQ.KHb = h->sKHb;
        const dim3 grid(BQ.R * BQ.C), blk(64 * BQ.WT);
        // saveat: which save indices each accepted attempt covers (same float comparisons as the forward controller)
        std::vector<int> sv_lo(n_att, 0), sv_hi(n_att, 0);
        if (!h->saveat.empty()) {
            int ns = (h->saveat[0] == h->t0) ? 1 : 0;
            for (int n = 0; n < n_att; ++n) {
                sv_lo[n] = ns;
                if (h->h_meta[n].flags & F_ACCEPT) {
                    const float tnew = h->h_meta[n].t + h->h_meta[n].dt;
                    while (ns < (int)h->saveat.size() && h->saveat[ns] <= tnew) ++ns;
                }
                sv_hi[n] = ns;
            }
        }
        // cotangent coefficients of eigen_est for an attempt (host-known: saveval cotangent, callback form, recorded norms)
        auto eig_coefs = [&](int n, float& c1, float& c2) {
            c1 = 0.f; c2 = 0.f;
            const StepMeta& mm = h->h_meta[n];
            const bool eg_ok = !(mm.eigen == 0.f || mm.eigen != mm.eigen);
            double eigb = 0.0;
            if (h->cfg.regularize == RNDE_REG_STIFF && eg_ok) eigb = (double)b.h_svb[n] * (mm.eigen > 0 ? 1.0 : -1.0) / 3.5068;
            if (h->cfg.regularize == RNDE_REG_ERR_STIFF && eg_ok) eigb = 0.1 * (double)b.h_svb[n] / 3.5068;
            if (eigb != 0.0 && mm.n1 > 0.f && mm.n2 > 0.f) {
                c1 = (float)(eigb / ((double)mm.n2 * (double)mm.n1));
                c2 = (float)(-eigb * ((double)mm.n1 / (double)mm.n2) / ((double)mm.n2 * (double)mm.n2));
            }
        };
        for (int n = n_att - 1; n >= 0; --n) {
            float c1 = 0.f, c2 = 0.f;
            eig_coefs(n, c1, c2);
            const double qo = pow((double)h->h_meta[n].qold_in, (double)kBeta2);   // for the scalar adjoint chain of the attempt
            if (h->persist == 1) {   // the attempt's 7 reverse launches as one (rnde_bstage_persist.h)
                PersistSync Y{h->tslab, h->pabort, h->pxcc, h->persist_spins};
                HIPCHK(h, slab_prepare(h, Q.F.Bpad, s));
                const dim3 pgrid(8 * BQ.R * ((BQ.C + 7) / 8));
                const bool fix = BQ.WT == 7 && BQ.HT == 7 && BQ.KHb == 7 && BQ.MT == 49 && BQ.R == 7 && h->D == 784 && h->H == 100 && !h->stage_generic;
                if (fix) {
                    // (+ the START-staged tape operands of the six stages, rnde_bstage_persist.h: 6 x 7 waves x 2 arrays x 1 KiB)
                    const size_t flds = h->stage_lds + (size_t)(RNDE_BSTAGE_HDMA ? 1 : 0) * 6 * 7 * 2 * 1024;
                    static const hipError_t attr = [&] {
                        hipError_t e = hipFuncSetAttribute((const void*)rnde_bstage_attempt_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        return e == hipSuccess ? hipFuncSetAttribute((const void*)rnde_bstage_attempt_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) : e;
                    }();
                    HIPCHK(h, attr);
                    if (h->act2) hipLaunchKernelGGL((rnde_bstage_attempt_kernel<1, 1>), pgrid, blk, flds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                    else hipLaunchKernelGGL((rnde_bstage_attempt_kernel<0, 1>), pgrid, blk, flds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                } else if (h->act2) hipLaunchKernelGGL((rnde_bstage_attempt_kernel<1, 0>), pgrid, blk, h->stage_lds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                else hipLaunchKernelGGL((rnde_bstage_attempt_kernel<0, 0>), pgrid, blk, h->stage_lds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                if ((st = couple_sum(h, b.bpart + (size_t)(n & 1) * Q.bpart_n * 4, 4LL * Q.bpart_n, s)) != RNDE_OK) return st;
                if (n >= n_att - side_att && (hi_att - n >= group || n == n_att - side_att)) {   // attempts [n, hi_att) are final
                    st = wgrad_group(2 + 6 * n, 2 + 6 * hi_att, true);
                    if (st != RNDE_OK) return st;
                    hi_att = n;
                }
                continue;
            }
            if (h->act2) hipLaunchKernelGGL((rnde_bstage_kernel<1, BM_START>), grid, blk, h->stage_lds, s, BQ, n, 0, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], qo);
            else hipLaunchKernelGGL((rnde_bstage_kernel<0, BM_START>), grid, blk, h->stage_lds, s, BQ, n, 0, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], qo);
            for (int j = 6; j >= 1; --j) {
                if (h->act2) hipLaunchKernelGGL((rnde_bstage_kernel<1, BM_STAGE>), grid, blk, h->stage_lds, s, BQ, n, j, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], qo);
                else hipLaunchKernelGGL((rnde_bstage_kernel<0, BM_STAGE>), grid, blk, h->stage_lds, s, BQ, n, j, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], qo);
            }
            if ((st = couple_sum(h, b.bpart + (size_t)(n & 1) * Q.bpart_n * 4, 4LL * Q.bpart_n, s)) != RNDE_OK) return st;
        }
        HIPCHK(h, hipGetLastError());
        {   // reverse of the initial-step rule: four stage-engine launches (rnde_binit_stage.h)
            if (h->act2) {
                hipLaunchKernelGGL((rnde_binit_stage_kernel<1, 0>), grid, blk, h->stage_lds, s, BQ);
                hipLaunchKernelGGL((rnde_binit_stage_kernel<1, 1>), grid, blk, h->stage_lds, s, BQ);
            } else {
                hipLaunchKernelGGL((rnde_binit_stage_kernel<0, 0>), grid, blk, h->stage_lds, s, BQ);
                hipLaunchKernelGGL((rnde_binit_stage_kernel<0, 1>), grid, blk, h->stage_lds, s, BQ);
            }
            if ((st = couple_sum(h, Q.ipart, 4LL * Q.F.nwg, s)) != RNDE_OK) return st;                       // (coupled controller: dot, tau of the reversed second evaluation)
            if (h->act2) {
                hipLaunchKernelGGL((rnde_binit_stage_kernel<1, 2>), grid, blk, h->stage_lds, s, BQ);
                hipLaunchKernelGGL((rnde_binit_stage_kernel<1, 3>), grid, blk, h->stage_lds, s, BQ);
            } else {
                hipLaunchKernelGGL((rnde_binit_stage_kernel<0, 2>), grid, blk, h->stage_lds, s, BQ);
                hipLaunchKernelGGL((rnde_binit_stage_kernel<0, 3>), grid, blk, h->stage_lds, s, BQ);
            }
            if ((st = couple_sum(h, Q.ipart + 4LL * Q.F.nwg, 4LL * Q.F.nwg, s)) != RNDE_OK) return st;      // tau of the first
            hipLaunchKernelGGL(rnde_bfin_kernel, dim3(1), dim3(64), 0, s, Q);
            HIPCHK(h, hipGetLastError());
        }
    }
    if (h->timing) HIPCHK(h, hipEventRecord(h->tev[3], s));
    // remaining evaluations on all CUs (everything that did not go to the side stream, incl. the two initialisation evaluations)
    st = wgrad_group(0, 2 + 6 * hi_att, false);
    if (st != RNDE_OK) return st;
    if (used_side) {
        hipEvent_t ev = h->wevents[64];
        HIPCHK(h, hipEventRecord(ev, h->wstream));
        HIPCHK(h, hipStreamWaitEvent(s, ev, 0));
    }
    if (cur1 > 16 && cur2 > 16) {   // both layers in one launch per pass (same sums in the same order as launch_wgrad_reduce)
        const long long len1 = (long long)h->H * (h->D + 2), len2 = (long long)h->D * (h->H + 2);
        const size_t seg_r = std::max((size_t)len1, (size_t)len2);
        float* r1 = b.slab_r; float* r2 = b.slab_r + 16 * seg_r;
        const int pg1 = (cur1 + 15) / 16, g1 = (cur1 + pg1 - 1) / pg1, pg2 = (cur2 + 15) / 16, g2 = (cur2 + pg2 - 1) / pg2;
        const int grid = (int)std::min<long long>((std::max(len1, len2) + 255) / 256, 2048);
        ReducePair A{{{slab1, r1, len1, cur1, pg1}, {slab2w, r2, len2, cur2, pg2}}};
        hipLaunchKernelGGL(rnde_wgrad_reduce_pair, dim3(grid, std::max(g1, g2), 2), dim3(256), 0, s, A);
        ReducePair Bp{{{r1, p_bar_dev, len1, g1, g1}, {r2, p_bar_dev + (size_t)h->H * (h->D + 2), len2, g2, g2}}};
        hipLaunchKernelGGL(rnde_wgrad_reduce_pair, dim3(grid, 1, 2), dim3(256), 0, s, Bp);
        HIPCHK(h, hipGetLastError());
    } else {
        st = launch_wgrad_reduce(h, slab1, cur1, h->H, h->D, p_bar_dev, s);                                    // [W1; b1]
        if (st != RNDE_OK) return st;
        st = launch_wgrad_reduce(h, slab2w, cur2, h->D, h->H, p_bar_dev + (size_t)h->H * (h->D + 2), s);      // [W2; b2]
        if (st != RNDE_OK) return st;
    }
    if (h->timing) { HIPCHK(h, hipEventRecord(h->tev[4], s)); h->tev_bwd = true; }
#ifdef RNDE_DIAG
    if (h->engine == 2 && h->persist == 1 && getenv("RNDE_DIAG_BWD")) {
        unsigned long long hst[64] = {0};
        hipStreamSynchronize(s);
        hipMemcpy(hst, h->diag_buf, sizeof(hst), hipMemcpyDeviceToHost);
        fprintf(stderr, "persistent reverse attempt (workgroup 0 thread 0, cycles): START %lld (entry->weights issued %lld, ->w1t %lld, ->array loads issued %lld, finish_attempt_scalars %lld, scalar chain %lld, vector part %lld), its phase D + put %lld\n", (long long)(hst[1]-hst[0]), (long long)(hst[43]-hst[0]), (long long)(hst[44]-hst[43]), (long long)(hst[40]-hst[44]), (long long)(hst[41]-hst[40]), (long long)(hst[42]-hst[41]), (long long)(hst[1]-hst[42]), (long long)(hst[2]-hst[1]));
        for (int st = 0; st < 6; ++st) { const unsigned long long* q = hst + 3 + 5 * st; const unsigned long long prev = st == 0 ? hst[2] : hst[7 + 5 * (st - 1)];
            fprintf(stderr, "  stage j=%d: poll %lld A %lld B %lld C %lld D+put %lld\n", 6 - st, (long long)(q[0]-prev), (long long)(q[1]-q[0]), (long long)(q[2]-q[1]), (long long)(q[3]-q[2]), st < 5 ? (long long)(q[4]-q[3]) : 0LL); }
        fprintf(stderr, "  total %lld cycles\n", (long long)(hst[34]-hst[0]));
    }
#endif
    if (!sync) {   // rnde_node_backward_async: no host round trip; the health words are looked at by the next synchronising call
        if (tspan_bar_dev) HIPCHK(h, hipMemcpyAsync(tspan_bar_dev, b.tspan_out, 8, hipMemcpyDeviceToDevice, s));
        h->have_tape = false;
        h->pending_bwd = (h->engine == 2 && h->persist == 1);
        return RNDE_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_scal, b.tspan_out, 8, hipMemcpyDeviceToHost, s));
    if (h->engine == 2) persist_check_enqueue(h, h->sR * (Q.F.Bpad / 16), s);
    HIPCHK(h, hipStreamSynchronize(s));
    if (h->couple && rnde_comm_health(h->couple) != RNDE_OK) { h->err = std::string("coupled controller: ") + rnde_comm_last_error(h->couple); return RNDE_ERR_HIP; }
    if (tspan_bar_host) { tspan_bar_host[0] = h->h_scal[0]; tspan_bar_host[1] = h->h_scal[1]; }
    h->have_tape = false;  // z2bar overwrote k_s in place: the tape is consumed
    if (h->engine == 2 && persist_check_result(h, Q.F.Bpad / 16, h->sR, s)) {
        h->err = "persistent reverse kernel abandoned its hand-off (tape consumed): rerun forward + backward, the multi-launch kernels are now in use";
        return RNDE_ERR_HIP;
    }
    return RNDE_OK;
}

// ---- fused classifier head ------------------------------------------------------
static rnde_status head_reserve(rnde_node* h, int32_t B, int32_t n_classes) {
    const size_t need = (size_t)B * n_classes + B + (size_t)kHeadChunks * n_classes * h->D;
    if (h->head_ws_floats < need) {
        if (h->head_ws) hipFree(h->head_ws);
        h->head_ws = nullptr; h->head_ws_floats = 0;
        HIPCHK(h, hipMalloc((void**)&h->head_ws, need * 4));
        h->head_ws_floats = need;
    }
    return RNDE_OK;
}
extern "C" rnde_status rnde_classifier_head(rnde_node* h, const float* u_dev, const float* p3_dev, const float* y_dev,
                                            int32_t B, int32_t n_classes, float* logits_out_dev, float* u_bar_dev,
                                            float* p3_bar_dev, float* ce_out_dev, void* stream) {
    if (!h || B < 1 || n_classes < 1 || n_classes > kHeadMaxC) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const rnde_status rs = head_reserve(h, B, n_classes);
    if (rs != RNDE_OK) return rs;
    float* delta = h->head_ws;
    float* ce_col = h->head_ws + (size_t)B * n_classes;
    if (h->D > 256 * kHeadRowsPerThread) { h->err = "classifier head: D <= 1024"; return RNDE_ERR_BAD_ARG; }
    if (n_classes == 10) hipLaunchKernelGGL((rnde_head_col_kernel<10>), dim3(B), dim3(256), 0, s, u_dev, p3_dev, y_dev, h->D, n_classes, B,
                       logits_out_dev, u_bar_dev, delta, ce_col);
    else hipLaunchKernelGGL((rnde_head_col_kernel<0>), dim3(B), dim3(256), 0, s, u_dev, p3_dev, y_dev, h->D, n_classes, B,
                       logits_out_dev, u_bar_dev, delta, ce_col);
    float* partial = ce_col + B;
    hipLaunchKernelGGL(rnde_head_wgrad_kernel, dim3((h->D + 255) / 256, kHeadChunks), dim3(256), 0, s, u_dev, (const float*)delta,
                       h->D, n_classes, B, partial);
    hipLaunchKernelGGL(rnde_head_reduce_kernel, dim3((n_classes * h->D + 255) / 256), dim3(256), 0, s, (const float*)partial,
                       (const float*)delta, (const float*)ce_col, h->D, n_classes, B, p3_bar_dev, ce_out_dev);
    HIPCHK(h, hipGetLastError());
    return RNDE_OK;
}

// ---- one training-step gradient in ONE call (forward solve -> head -> reverse solve), SURVEY.md 8f rank 1 ------------------
// The three calls above chained by the caller leave the GPU idle between the solve and its reverse (~80 us of a 2.4 ms step at
// B = 512): the forward ends in a host wait (the host needs the step log to launch the reverse sweep), and only then does the
// caller queue the head and the reverse pass.  Here the head and the weight packs of the reverse sweep are queued BEFORE that
// wait (they do not depend on the step log), so they run while the host wakes up and prepares the sweep.
extern "C" rnde_status rnde_node_classifier_grad(rnde_node* h, const float* x_dev, const float* p2_dev, const float* p3_dev,
                                                 const float* y_dev, int32_t B, int32_t n_classes, float t0, float t1,
                                                 float lambda, float* p2_bar_dev, float* p3_bar_dev, float* x_bar_dev,
                                                 float* ce_out_dev, float* reg_out_host, int64_t* nfe_out, rnde_comm* comm,
                                                 void* stream) {
    if (!h || !x_dev || !p2_dev || !p3_dev || !y_dev || !p2_bar_dev || !p3_bar_dev || !ce_out_dev) return RNDE_ERR_BAD_ARG;
    if (h->engine != 2) { h->err = "rnde_node_classifier_grad: two-layer dynamics on the stage engine (col_tile 0)"; return RNDE_ERR_BAD_ARG; }
    if (B < 1 || B > h->cfg.max_batch) { h->err = "bad B or tspan"; return RNDE_ERR_BAD_ARG; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t A = (size_t)h->D * B;
    if (h->cg_ws_floats < 3 * A) {
        if (h->cg_ws) hipFree(h->cg_ws);
        h->cg_ws = nullptr; h->cg_ws_floats = 0;
        HIPCHK(h, hipMalloc((void**)&h->cg_ws, 3 * A * 4));
        h->cg_ws_floats = 3 * A;
    }
    if (!h->ev_host) HIPCHK(h, hipEventCreateWithFlags(&h->ev_host, hipEventDisableTiming));
    if (n_classes < 1 || n_classes > kHeadMaxC) return RNDE_ERR_BAD_ARG;
    rnde_status st = head_reserve(h, B, n_classes);   // (the hook below runs between an event record and the host's wait on it: it only enqueues)
    if (st != RNDE_OK) return st;
    float* u = h->cg_ws; float* ubar = h->cg_ws + A; float* xbar = x_bar_dev ? x_bar_dev : h->cg_ws + 2 * A;
    h->cg_sv.resize((size_t)h->cfg.max_attempts + 1);
    int32_t nsv = 0;
    int64_t nfe = 0;
    h->after_solve = [&](hipStream_t s) -> rnde_status {
        const rnde_status r = rnde_classifier_head(h, u, p3_dev, y_dev, B, n_classes, nullptr, ubar, p3_bar_dev, ce_out_dev, s);
        return r;
    };
    st = forward_impl(h, x_dev, p2_dev, B, t0, t1, u, nullptr, 0, nullptr, &nfe, h->cg_sv.data(), &nsv, 1, stream);
    h->after_solve = nullptr;
    if (st != RNDE_OK) { h->rev_packed = false; return st; }
    // lambda * mean(sv.saveval) (experiments/mnist_node.jl:135): every saved value carries the cotangent lambda / n
    double reg = 0.0;
    const bool regularize = lambda != 0.f && nsv > 0 && h->cfg.regularize != RNDE_REG_NONE;
    if (regularize) {
        for (int i = 0; i < nsv; ++i) reg += h->cg_sv[i];
        reg = (double)lambda * reg / nsv;
        for (int i = 0; i < nsv; ++i) h->cg_sv[i] = lambda / (float)nsv;
    }
    if (reg_out_host) *reg_out_host = (float)reg;
    if (nfe_out) *nfe_out = nfe;
    const int64_t n3 = (int64_t)n_classes * h->D + n_classes;
    st = bwd_run(h, ubar, regularize ? h->cg_sv.data() : nullptr, xbar, p2_bar_dev, nullptr, (hipStream_t)stream, false, nullptr);
    h->rev_packed = false;
    if (st != RNDE_OK) return st;
    if (comm) {   // ONE collective per step when the two gradients sit back to back ([p2-bar | p3-bar], the flat buffer of a data-parallel caller):
        // on the caller's stream everything is serial anyway, and a second call is a second RCCL launch latency
        if (p3_bar_dev == p2_bar_dev + h->P) st = rnde_comm_allreduce(comm, p2_bar_dev, (int64_t)h->P + n3, 0, stream);
        else if ((st = rnde_comm_allreduce(comm, p2_bar_dev, (int64_t)h->P, 0, stream)) == RNDE_OK) st = rnde_comm_allreduce(comm, p3_bar_dev, n3, 0, stream);
        if (st != RNDE_OK) { h->err = std::string("all-reduce: ") + rnde_comm_last_error(comm); return st; }
    }
    return RNDE_OK;
}

// ---- chain engine reverse pass ----------------------------------------------------------------------------
template <int NKD, int ALT = 0>
static hipError_t launch_bchain_t(rnde_node* h, const BChainParams& Q, const std::vector<int>& sv_lo, const std::vector<int>& sv_hi, hipStream_t s) {
    const BwdBuffers& b = h->bw;
    const size_t lds = h->chain_lds_b;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_bchain_kernel<NKD, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchain_init_kernel<NKD, 1, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchain_init_kernel<NKD, 2, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 grid(Q.B.F.nwg), blk(64 * kCW);
    for (int n = Q.B.n_att - 1; n >= 0; --n) {
        float c1 = 0.f, c2 = 0.f;   // cotangent of eigen_est for this attempt (as in bwd_run)
        const StepMeta& mm = h->h_meta[n];
        const bool eg_ok = !(mm.eigen == 0.f || mm.eigen != mm.eigen);
        double eigb = 0.0;
        if (h->cfg.regularize == RNDE_REG_STIFF && eg_ok) eigb = (double)b.h_svb[n] * (mm.eigen > 0 ? 1.0 : -1.0) / 3.5068;
        if (h->cfg.regularize == RNDE_REG_ERR_STIFF && eg_ok) eigb = 0.1 * (double)b.h_svb[n] / 3.5068;
        if (eigb != 0.0 && mm.n1 > 0.f && mm.n2 > 0.f) {
            c1 = (float)(eigb / ((double)mm.n2 * (double)mm.n1));
            c2 = (float)(-eigb * ((double)mm.n1 / (double)mm.n2) / ((double)mm.n2 * (double)mm.n2));
        }
        hipLaunchKernelGGL((rnde_bchain_kernel<NKD, ALT>), grid, blk, lds, s, Q, n, mm, sv_lo[n], sv_hi[n], c1, c2);
    }
    hipLaunchKernelGGL((rnde_bchain_init_kernel<NKD, 1, ALT>), grid, blk, lds, s, Q);
    hipLaunchKernelGGL((rnde_bchain_init_kernel<NKD, 2, ALT>), grid, blk, lds, s, Q);
    hipLaunchKernelGGL(rnde_bfin_kernel, dim3(1), dim3(64), 0, s, Q.B);
    return hipGetLastError();
}


// ---- chain engine, multi-wave kernels: reverse pass (rnde_bchainmw.h) ------------------------------------------------------
template <int NR, int TAB, int LAT = 0>
static rnde_status launch_bmw_t(rnde_node* h, const BMwParams& Q, const std::vector<int>& sv_lo, const std::vector<int>& sv_hi, hipStream_t s) {
    const BwdBuffers& b = h->bw;
    const size_t lds = h->mw_lds_b;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_bchainmw_kernel<NR, TAB, LAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_kernel<NR, TAB, LAT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_init_kernel<NR, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_init_kernel<NR, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        HIPCHK(h, e);
        attr_set = true;
    }
    const dim3 grid(Q.ntiles), blk(kMwThreads);
    rnde_status st = RNDE_OK;
    // the whole sweep as ONE launch (rnde_bchainmw.h SWEEP): every workgroup resident (<= 256 column tiles; more than 32: meeting through the
    // memory side, as the forward solve), no shared controller, more than one attempt
    const bool sweep = h->mw_bsweep > 0 && !h->couple && Q.ntiles <= kMwMeetMax && Q.B.n_att >= 2 && h->mw_xch;
    int* a_lo = h->h_mw_bargs; int* a_hi = a_lo + h->cfg.max_attempts; float* a_eig = (float*)(a_hi + h->cfg.max_attempts);
    for (int n = Q.B.n_att - 1; n >= 0; --n) {
        float c1 = 0.f, c2 = 0.f;   // cotangent of eigen_est for this attempt (as in bwd_run)
        const StepMeta& mm = h->h_meta[n];
        const bool eg_ok = !(mm.eigen == 0.f || mm.eigen != mm.eigen);
        double eigb = 0.0;
        if (h->cfg.regularize == RNDE_REG_STIFF && eg_ok) eigb = (double)b.h_svb[n] * (mm.eigen > 0 ? 1.0 : -1.0) / 3.5068;
        if (h->cfg.regularize == RNDE_REG_ERR_STIFF && eg_ok) eigb = 0.1 * (double)b.h_svb[n] / 3.5068;
        if (eigb != 0.0 && mm.n1 > 0.f && mm.n2 > 0.f) {
            c1 = (float)(eigb / ((double)mm.n2 * (double)mm.n1));
            c2 = (float)(-eigb * ((double)mm.n1 / (double)mm.n2) / ((double)mm.n2 * (double)mm.n2));
        }
        if (sweep) { a_lo[n] = sv_lo[n]; a_hi[n] = sv_hi[n]; a_eig[2 * n] = c1; a_eig[2 * n + 1] = c2; continue; }
        hipLaunchKernelGGL((rnde_bchainmw_kernel<NR, TAB, LAT>), grid, blk, lds, s, Q, n, mm, sv_lo[n], sv_hi[n], c1, c2);
        // (coupled controller, SURVEY 8e mode 2: the S, tau, c-tau partials of attempt n summed over the ranks before attempt n - 1 reads them)
        if ((st = couple_sum(h, b.bpart + (size_t)(n & 1) * Q.B.bpart_n * 4, 4LL * Q.B.bpart_n, s)) != RNDE_OK) return st;
    }
    if (sweep) {
        const int cap = h->cfg.max_attempts;
        HIPCHK(h, hipMemcpyAsync(h->mw_bargs, h->h_mw_bargs, (size_t)cap * 16, hipMemcpyHostToDevice, s));
        if (++h->mw_epoch >= 500000u) { h->mw_epoch = 1; HIPCHK(h, hipMemsetAsync(h->mw_xch, 0, (size_t)(cap + 4) * 3 * kMwMeetMax * 8, s)); }
        BMwParams W = Q;
        W.sv_lo = h->mw_bargs; W.sv_hi = h->mw_bargs + cap; W.eig_c = (const float*)(h->mw_bargs + 2 * cap);
        W.xch = h->mw_xch; W.xcc = h->mw_xcc; W.abort_word = h->mw_abort; W.epoch = h->mw_epoch; W.xch_global = Q.ntiles > 32 ? 1 : 0; W.xcd_slot = h->mw_slot;
        hipLaunchKernelGGL((rnde_bchainmw_kernel<NR, TAB, LAT, 1>), dim3(W.xch_global ? Q.ntiles : 8 * Q.ntiles), blk, lds, s, W, Q.B.n_att - 1, h->h_meta[Q.B.n_att - 1], 0, 0, 0.f, 0.f);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(h->h_mw_bchk, h->mw_abort, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->h_mw_bchk + 2, h->mw_xcc, (size_t)Q.ntiles * 4, hipMemcpyDeviceToHost, s));
        h->pending_bsweep = true;
    }
    hipLaunchKernelGGL((rnde_bchainmw_init_kernel<NR, 1>), grid, blk, lds, s, Q);
    if ((st = couple_sum(h, Q.B.ipart, 4LL * Q.B.F.nwg, s)) != RNDE_OK) return st;                        // dot, tau of the reversed second evaluation
    hipLaunchKernelGGL((rnde_bchainmw_init_kernel<NR, 2>), grid, blk, lds, s, Q);
    if ((st = couple_sum(h, Q.B.ipart + 4LL * Q.B.F.nwg, 4LL * Q.B.F.nwg, s)) != RNDE_OK) return st;      // tau of the first
    hipLaunchKernelGGL(rnde_bfin_kernel, dim3(1), dim3(64), 0, s, Q.B);
    HIPCHK(h, hipGetLastError());
    return RNDE_OK;
}

static rnde_status chain_mw_bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                    float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync, float* tspan_bar_dev) {
    BwdBuffers& b = h->bw;
    const ChainGeo& G = h->cg;
    const int cap = h->cfg.max_attempts, ntiles_max = h->Bpad_max / 16;
    const int E = h->rk_S - 1;      // evaluations per attempted step (6; S - 1 for an S-stage table)
    if (!b.ready) {
        const size_t Ac = (size_t)ntiles_max * h->NKD * 64;
        HIPCHK(h, hipMalloc((void**)&b.U, Ac * 4)); HIPCHK(h, hipMalloc((void**)&b.K1, Ac * 4)); HIPCHK(h, hipMalloc((void**)&b.UB1, Ac * 4));
        HIPCHK(h, hipMalloc((void**)&b.svb_att, (size_t)cap * 4));
        HIPCHK(h, hipMalloc((void**)&b.bstate, 2 * sizeof(BState))); HIPCHK(h, hipMalloc((void**)&b.ibstate, 2 * sizeof(IBState)));
        HIPCHK(h, hipMalloc((void**)&b.bpart, (size_t)(2 * h->nwg_max + 256) * 4 * 4)); HIPCHK(h, hipMalloc((void**)&b.ipart, (size_t)2 * h->nwg_max * 4 * 4));   // (+256 entries: finish_attempt_scalars reads whole 256-entry blocks)
        HIPCHK(h, hipMalloc((void**)&b.tspan_out, 2 * 4));
        HIPCHK(h, hipHostMalloc((void**)&b.h_svb, (size_t)cap * 4));
        HIPCHK(h, hipMalloc((void**)&b.slab, (size_t)96 * h->P * 4)); HIPCHK(h, hipMalloc((void**)&b.slab_r, (size_t)16 * h->P * 4));
        HIPCHK(h, hipMalloc((void**)&h->ev_t, ((size_t)E * cap + 2) * 4)); HIPCHK(h, hipHostMalloc((void**)&h->h_ev_t, ((size_t)E * cap + 2) * 4));
        b.ready = true;
    }
    const int n_att = h->n_att, n_evals = E * n_att + 2;
    if (!h->mw_slab || h->mw_slab_evals < n_evals) { h->err = "activation slab missing: the forward was not taped on the multi-wave kernels"; return RNDE_ERR_NO_TAPE; }
    // (coupled controller: every rank passes the cotangent of its own loss; the shared scalars carry `world` times it -- as in bwd_run)
    const float svb_scale = h->couple ? (float)h->couple_world : 1.f;
    for (int i = 0; i < n_att; ++i)
        b.h_svb[i] = (saveval_bar_host && h->sv_index[i] >= 0) ? svb_scale * saveval_bar_host[h->sv_index[i]] : 0.f;
    HIPCHK(h, hipMemcpyAsync(b.svb_att, b.h_svb, (size_t)std::max(1, n_att) * 4, hipMemcpyHostToDevice, s));
    BMwParams Q{};
    Q.B.F = make_params(h, h->xcopy, h->B, h->t0, h->t1, 1);
    Q.B.U = b.U; Q.B.K1 = b.K1; Q.B.UB1 = b.UB1; Q.B.svb_att = b.svb_att;
    Q.B.bstate = b.bstate; Q.B.ibstate = b.ibstate; Q.B.bpart = b.bpart; Q.B.ipart = b.ipart;
    Q.B.ubar = u_bar_dev; Q.B.xbar = x_bar_dev; Q.B.tspan_out = b.tspan_out;
    Q.B.n_att = n_att; Q.B.track_ctrl = h->cfg.track_ctrl; Q.B.track_initdt = h->cfg.track_initdt; Q.B.reg_kind = h->cfg.regularize;
    Q.B.bpart_n = Q.B.F.nwg;
    Q.B.tspan_scale = h->couple ? 1.f / (float)h->couple_world : 1.f;
    Q.B.sv_T = (int)h->saveat.size();
    Q.B.sv_ubar0 = (!h->saveat.empty() && h->saveat[0] == h->t0) ? u_bar_dev : nullptr;
    Q.G = h->mg; Q.rk = h->rk; Q.tab = h->mw_tab; Q.ntiles = Q.B.F.Bpad / 16;
    Q.slab = h->mw_slab; Q.ev_stride = (long long)Q.ntiles * h->mg.RS * 64;
    Q.sv_t = h->saveat.empty() ? nullptr : h->sv_t_dev; Q.sv_ubar = u_bar_dev; Q.nsave = (int)h->saveat.size();
    // evaluation times in slab order (0: f(u0,t0), 1: f(u1,t0+dt0), 2 + 6n + (s-1): stage s of attempt n), save indices per accepted attempt
    std::vector<int> sv_lo(std::max(1, n_att), 0), sv_hi(std::max(1, n_att), 0);
    {
        int ns = (!h->saveat.empty() && h->saveat[0] == h->t0) ? 1 : 0;
        h->h_ev_t[0] = h->t0; h->h_ev_t[1] = h->t0 + h->h_init->dt0;
        for (int n = 0; n < n_att; ++n) {
            const StepMeta& m = h->h_meta[n];
            for (int sidx = 2; sidx <= h->rk_S; ++sidx) h->h_ev_t[2 + E * n + sidx - 2] = m.t + (h->rk_tab ? h->rk.c[sidx - 1] : tsC(sidx - 1)) * m.dt;
            sv_lo[n] = ns;
            if (m.flags & F_ACCEPT) {
                const float tnew = m.t + m.dt;
                while (ns < (int)h->saveat.size() && h->saveat[ns] <= tnew) ++ns;
            }
            sv_hi[n] = ns;
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->ev_t, h->h_ev_t, (size_t)n_evals * 4, hipMemcpyHostToDevice, s));
    rnde_status e;
    if (h->rk_tab == 2) e = h->NKD == 4 ? launch_bmw_t<1, 2>(h, Q, sv_lo, sv_hi, s) : (h->NKD == 8 ? launch_bmw_t<2, 2>(h, Q, sv_lo, sv_hi, s) : launch_bmw_t<4, 2>(h, Q, sv_lo, sv_hi, s));
    else if (h->mw_lat) e = h->rk_tab ? launch_bmw_t<2, 1, 1>(h, Q, sv_lo, sv_hi, s) : launch_bmw_t<2, 0, 1>(h, Q, sv_lo, sv_hi, s);   // latent-ODE shape: transposed weights register stationary
    else if (h->rk_tab) e = h->NKD == 4 ? launch_bmw_t<1, 1>(h, Q, sv_lo, sv_hi, s) : (h->NKD == 8 ? launch_bmw_t<2, 1>(h, Q, sv_lo, sv_hi, s) : launch_bmw_t<4, 1>(h, Q, sv_lo, sv_hi, s));
    else e = h->NKD == 4 ? launch_bmw_t<1, 0>(h, Q, sv_lo, sv_hi, s) : (h->NKD == 8 ? launch_bmw_t<2, 0>(h, Q, sv_lo, sv_hi, s) : launch_bmw_t<4, 0>(h, Q, sv_lo, sv_hi, s));
    if (e != RNDE_OK) return e;
    // parameter gradients of all layers over all evaluations: the one-wave engine's kernel on the same slab format
    BChainParams W{};
    W.G = G; W.ntiles = Q.ntiles; W.slab = h->mw_slab; W.ev_stride = Q.ev_stride; W.RS = h->mg.RS;
    for (int l = 0; l <= G.n_layers; ++l) { W.hrow[l] = h->mg.hrow[l]; if (l < G.n_layers) W.zrow[l] = h->mg.zrow[l]; }
    const int n_units = n_evals * Q.ntiles;
    const int chunks = std::max(1, std::min(96, n_units / 8));
    const int per_chunk = (n_units + chunks - 1) / chunks;
    hipLaunchKernelGGL(rnde_chain_wgrad_kernel, dim3(G.n_layers, chunks), dim3(64 * kCW), 0, s, W, (const float*)h->ev_t, n_units, per_chunk, b.slab, h->P);
    HIPCHK(h, hipGetLastError());
    {
        const long long len = h->P;
        const int grid = (int)std::min<long long>((len + 255) / 256, 2048);
        if (chunks <= 16) hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)b.slab, chunks, chunks, len, p_bar_dev);
        else {
            const int per_group = (chunks + 15) / 16, groups = (chunks + per_group - 1) / per_group;
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, groups), dim3(256), 0, s, (const float*)b.slab, chunks, per_group, len, b.slab_r);
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)b.slab_r, groups, groups, len, p_bar_dev);
        }
        HIPCHK(h, hipGetLastError());
    }
    h->have_tape = false;
    if (!sync) {
        if (tspan_bar_dev) HIPCHK(h, hipMemcpyAsync(tspan_bar_dev, b.tspan_out, 8, hipMemcpyDeviceToDevice, s));
        return RNDE_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_scal, b.tspan_out, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (bsweep_failed(h, s))     // nothing the sweep reads was consumed: the same reverse pass again, one launch per attempt
        return chain_mw_bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, tspan_bar_host, s, sync, tspan_bar_dev);
    if (tspan_bar_host) { tspan_bar_host[0] = h->h_scal[0]; tspan_bar_host[1] = h->h_scal[1]; }
    return RNDE_OK;
}

static rnde_status chain_bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                 float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync, float* tspan_bar_dev) {
    if (h->mw) { HIPCHK(h, hipSetDevice(h->cfg.device)); return chain_mw_bwd_run(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, tspan_bar_host, s, sync, tspan_bar_dev); }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    BwdBuffers& b = h->bw;
    const ChainGeo& G = h->cg;
    const int cap = h->cfg.max_attempts, ntiles_max = h->Bpad_max / 16;
    if (!b.ready) {
        const size_t Ac = (size_t)ntiles_max * h->NKD * 64;
        HIPCHK(h, hipMalloc((void**)&b.U, Ac * 4)); HIPCHK(h, hipMalloc((void**)&b.K1, Ac * 4)); HIPCHK(h, hipMalloc((void**)&b.UB1, Ac * 4));
        HIPCHK(h, hipMalloc((void**)&b.svb_att, (size_t)cap * 4));
        HIPCHK(h, hipMalloc((void**)&b.bstate, 2 * sizeof(BState))); HIPCHK(h, hipMalloc((void**)&b.ibstate, 2 * sizeof(IBState)));
        HIPCHK(h, hipMalloc((void**)&b.bpart, (size_t)(2 * h->nwg_max + 256) * 4 * 4)); HIPCHK(h, hipMalloc((void**)&b.ipart, (size_t)2 * h->nwg_max * 4 * 4));   // (+256 entries: finish_attempt_scalars reads whole 256-entry blocks)
        HIPCHK(h, hipMalloc((void**)&b.tspan_out, 2 * 4));
        HIPCHK(h, hipHostMalloc((void**)&b.h_svb, (size_t)cap * 4));
        HIPCHK(h, hipMalloc((void**)&b.slab, (size_t)96 * h->P * 4)); HIPCHK(h, hipMalloc((void**)&b.slab_r, (size_t)16 * h->P * 4));
        HIPCHK(h, hipMalloc((void**)&h->ev_t, ((size_t)6 * cap + 2) * 4)); HIPCHK(h, hipHostMalloc((void**)&h->h_ev_t, ((size_t)6 * cap + 2) * 4));
        b.ready = true;
    }
    const int n_att = h->n_att, n_evals = 6 * n_att + 2;
    for (int i = 0; i < n_att; ++i)
        b.h_svb[i] = (saveval_bar_host && h->sv_index[i] >= 0) ? saveval_bar_host[h->sv_index[i]] : 0.f;
    HIPCHK(h, hipMemcpyAsync(b.svb_att, b.h_svb, (size_t)std::max(1, n_att) * 4, hipMemcpyHostToDevice, s));
    BChainParams Q{};
    Q.B.F = make_params(h, h->xcopy, h->B, h->t0, h->t1, 1);
    Q.B.U = b.U; Q.B.K1 = b.K1; Q.B.UB1 = b.UB1; Q.B.svb_att = b.svb_att;
    Q.B.bstate = b.bstate; Q.B.ibstate = b.ibstate; Q.B.bpart = b.bpart; Q.B.ipart = b.ipart;
    Q.B.ubar = u_bar_dev; Q.B.xbar = x_bar_dev; Q.B.tspan_out = b.tspan_out;
    Q.B.n_att = n_att; Q.B.track_ctrl = h->cfg.track_ctrl; Q.B.track_initdt = h->cfg.track_initdt; Q.B.reg_kind = h->cfg.regularize;
    Q.B.bpart_n = Q.B.F.nwg;
    Q.B.tspan_scale = h->couple ? 1.f / (float)h->couple_world : 1.f;
    Q.B.sv_T = (int)h->saveat.size();
    Q.B.sv_ubar0 = (!h->saveat.empty() && h->saveat[0] == h->t0) ? u_bar_dev : nullptr;
    Q.G = G; Q.frags = h->cfrags; Q.ntiles = Q.B.F.Bpad / 16;
    int row = 0;
    auto pad4 = [](int k) { return 4 * ((k + 3) / 4); };
    for (int l = 0; l < G.n_layers; ++l) { Q.hrow[l] = row; row += pad4(G.nks[l]); Q.zrow[l] = row; row += pad4(G.nks[l + 1]); }
    Q.hrow[G.n_layers] = row; row += pad4(G.nks[G.n_layers]);
    Q.RS = row; Q.ev_stride = (long long)Q.ntiles * row * 64;
    Q.sv_t = h->saveat.empty() ? nullptr : h->sv_t_dev; Q.sv_ubar = u_bar_dev; Q.nsave = (int)h->saveat.size();
    const size_t need = (size_t)n_evals * Q.ev_stride;
    if (h->cslab_floats < need) {
        if (h->cslab) hipFree(h->cslab);
        h->cslab = nullptr; h->cslab_floats = 0;
        const size_t grow = need + need / 2;        // (head room: a step count that creeps up while a model trains must not free + allocate every step)
        if (hipMalloc((void**)&h->cslab, grow * 4) == hipSuccess) h->cslab_floats = grow;
        else { HIPCHK(h, hipMalloc((void**)&h->cslab, need * 4)); h->cslab_floats = need; }
    }
    Q.slab = h->cslab;
    // evaluation times (time column of TDChain layers) and the save indices each accepted attempt covers
    std::vector<int> sv_lo(std::max(1, n_att), 0), sv_hi(std::max(1, n_att), 0);
    {
        int ns = (!h->saveat.empty() && h->saveat[0] == h->t0) ? 1 : 0;
        for (int n = 0; n < n_att; ++n) {
            const StepMeta& m = h->h_meta[n];
            for (int sidx = 2; sidx <= 7; ++sidx) h->h_ev_t[6 * n + sidx - 2] = m.t + tsC(sidx - 1) * m.dt;
            sv_lo[n] = ns;
            if (m.flags & F_ACCEPT) {
                const float tnew = m.t + m.dt;
                while (ns < (int)h->saveat.size() && h->saveat[ns] <= tnew) ++ns;
            }
            sv_hi[n] = ns;
        }
        h->h_ev_t[6 * n_att] = h->t0; h->h_ev_t[6 * n_att + 1] = h->t0 + h->h_init->dt0;
    }
    HIPCHK(h, hipMemcpyAsync(h->ev_t, h->h_ev_t, (size_t)n_evals * 4, hipMemcpyHostToDevice, s));
    hipError_t e;
    switch (h->NKD) {
        case 4: e = launch_bchain_t<4>(h, Q, sv_lo, sv_hi, s); break;
        case 8: e = h->chain_alt ? launch_bchain_t<8, 1>(h, Q, sv_lo, sv_hi, s) : launch_bchain_t<8>(h, Q, sv_lo, sv_hi, s); break;
        default: e = launch_bchain_t<16>(h, Q, sv_lo, sv_hi, s); break;
    }
    HIPCHK(h, e);
    // parameter gradients of all layers over all evaluations
    const int n_units = n_evals * Q.ntiles;
    const int chunks = std::max(1, std::min(96, n_units / 8));
    const int per_chunk = (n_units + chunks - 1) / chunks;
    hipLaunchKernelGGL(rnde_chain_wgrad_kernel, dim3(G.n_layers, chunks), dim3(64 * kCW), 0, s, Q, (const float*)h->ev_t, n_units, per_chunk, b.slab, h->P);
    HIPCHK(h, hipGetLastError());
    {
        const long long len = h->P;
        const int grid = (int)std::min<long long>((len + 255) / 256, 2048);
        if (chunks <= 16) hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)b.slab, chunks, chunks, len, p_bar_dev);
        else {
            const int per_group = (chunks + 15) / 16, groups = (chunks + per_group - 1) / per_group;
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, groups), dim3(256), 0, s, (const float*)b.slab, chunks, per_group, len, b.slab_r);
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)b.slab_r, groups, groups, len, p_bar_dev);
        }
        HIPCHK(h, hipGetLastError());
    }
    h->have_tape = false;
    if (!sync) {
        if (tspan_bar_dev) HIPCHK(h, hipMemcpyAsync(tspan_bar_dev, b.tspan_out, 8, hipMemcpyDeviceToDevice, s));
        return RNDE_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_scal, b.tspan_out, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (tspan_bar_host) { tspan_bar_host[0] = h->h_scal[0]; tspan_bar_host[1] = h->h_scal[1]; }
    return RNDE_OK;
}

extern "C" rnde_status rnde_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t len, int64_t t, float eta, float beta1,
                                      float beta2, float eps, float gscale, void* stream) {
    if (!p_dev || !g_dev || !m_dev || !v_dev || len < 0 || t < 1) return RNDE_ERR_BAD_ARG;
    if (len == 0) return RNDE_OK;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)t)), bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    hipLaunchKernelGGL(rnde::rnde_adam_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p_dev, g_dev, m_dev, v_dev,
                       (long long)len, gscale, eta, beta1, beta2, bc1, bc2, eps);
    return hipGetLastError() == hipSuccess ? RNDE_OK : RNDE_ERR_HIP;
}
extern "C" rnde_status rnde_momentum_step_scaled(float* p_dev, const float* g_dev, float* v_dev, int64_t len, int64_t n, float gamma,
                                                 float eta, float rho, float gscale, void* stream);
extern "C" rnde_status rnde_momentum_step(float* p_dev, const float* g_dev, float* v_dev, int64_t len, int64_t n, float gamma,
                                          float eta, float rho, void* stream) {
    return rnde_momentum_step_scaled(p_dev, g_dev, v_dev, len, n, gamma, eta, rho, 1.0f, stream);
}
extern "C" rnde_status rnde_momentum_step_scaled(float* p_dev, const float* g_dev, float* v_dev, int64_t len, int64_t n, float gamma,
                                                 float eta, float rho, float gscale, void* stream) {
    if (!p_dev || !g_dev || !v_dev || len < 0 || n < 1) return RNDE_ERR_BAD_ARG;
    if (len == 0) return RNDE_OK;
    const float inv_decay = gscale / (1.0f + gamma * (float)n);
    hipLaunchKernelGGL(rnde::rnde_momentum_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p_dev, g_dev,
                       v_dev, (long long)len, inv_decay, eta, rho);
    return hipGetLastError() == hipSuccess ? RNDE_OK : RNDE_ERR_HIP;
}
