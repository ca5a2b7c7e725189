// rnde_sde.hip -- C ABI of the stochastic layer (include/rnde.h: rnde_nsde_*) over the kernels of rnde_sde.h.
// No torch, no oracle, no CPU fallback.
#include "../../include/rnde.h"
// The kernel headers define (non-template) kernels in namespace rnde; rnde.hip includes them too.  This translation unit
// gets its own copy under another namespace name so that the two objects link into one library.
#define rnde rnde_sde_tu
#include "rnde_sde.h"
#include "rnde_sdemw.h"
#include "rnde_head.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace rnde;

struct rnde_nsde {
    rnde_nsde_config cfg{};
    int D = 0, Pf = 0, Pg = 0, P = 0, NKD = 8, Bpad_max = 0, ntiles_max = 0, nwg_max = 0;
    float* head_ws = nullptr; size_t head_ws_floats = 0;   // rnde_nsde_classifier_head
    // rnde_nsde_classifier_grad: work the forward enqueues behind the copies its host wait needs (it runs while the host wakes up), the
    // event that wait uses, and the buffers of the fused step
    std::function<rnde_status(hipStream_t)> after_solve; hipEvent_t ev_host = nullptr;
    float* cg_ws = nullptr; size_t cg_ws_floats = 0; std::vector<float> cg_sv;
    int mw = 0;    // 1: the four-waves-per-tile solve kernel (rnde_sdemw.h) for that shape, while a tile per workgroup still fits the chip
    size_t lds_mw = 0;
    int xch_wg = 0; // workgroups the exchange array is sized for
    int xch_local = 1;      // the four-waves-per-tile solve pins its workgroups to one XCD while they fit it (<= 32 tiles) and meets through that L2; 0 after the placement check failed once
    unsigned *xcc = nullptr, *h_xcc = nullptr;
    int pool_pred = 0;      // library noise: draws the next solve's pool is filled with (0 = all of max_attempts + 1); grows back on demand
    int fix = 0;   // 1: the reference's own shape (drift 8 -> 16 -> 8 k-steps, one-layer diffusion): kernels with compile-time shapes
    ChainGeo Gf{}, Gg{};
    SriTableau T{};
    float order = 1.5f, beta1 = 0, beta2 = 0, gamma = 0, qmin = 0, qmax = 0, qoldinit = 0, delta = 0;
    float *frags_f = nullptr, *frags_g = nullptr, *slots = nullptr, *tape = nullptr, *noise = nullptr, *replay = nullptr;
    size_t noise_floats = 0, tape_floats = 0;
    int n_slots = 0;
    SdeMeta *meta = nullptr, *h_meta = nullptr, *acc_meta = nullptr, *h_acc_meta = nullptr;
    SdeFinal *fin = nullptr, *h_fin = nullptr;
    unsigned long long* xch = nullptr;
    unsigned* abort_word = nullptr;
    float* eigpart = nullptr;      // RNDE_REG_STIFF: [max_attempts][2][workgroups]
    float *svb = nullptr, *h_svb = nullptr, *slab_f = nullptr, *slab_g = nullptr, *wslab = nullptr, *wslab_r = nullptr, *ev_t = nullptr;
    size_t slab_f_floats = 0, slab_g_floats = 0, ev_t_n = 0;
    float *part = nullptr, *h_part = nullptr;
    float* sv_t_dev = nullptr; size_t sv_cap = 0; std::vector<float> saveat;   // saveat times of the last forward ({R,true} methods)
    unsigned epoch = 0;
    size_t lds_fwd = 0, lds_bwd = 0;
    // last forward
    int B = 0, ntiles = 0, nwg = 0, n_att = 0, n_acc = 0, n_draws = 0, n_saveval = 0;
    float t0 = 0.f;
    bool have_tape = false;
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr}; bool tev_f = false, tev_b = false;   // around the solve kernel / the reverse sweep kernel
    std::vector<int> sv_index;   // per accepted step: index into saveval
    std::string err;
};

static thread_local std::string g_nsde_create_err;

#define SCHK(h, call)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);               \
            return RNDE_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

extern "C" const char* rnde_nsde_last_error(const rnde_nsde* h) { return h ? h->err.c_str() : g_nsde_create_err.c_str(); }

static int chain_params(int n_layers, const int32_t* dims) {
    int n = 0;
    for (int l = 0; l < n_layers; ++l) n += dims[l] * dims[l + 1] + dims[l + 1];
    return n;
}
extern "C" int32_t rnde_nsde_param_count(const rnde_nsde_config* c, int32_t* len_drift_out) {
    const int a = chain_params(c->drift_layers, c->drift_dims), b = chain_params(c->diff_layers, c->diff_dims);
    if (len_drift_out) *len_drift_out = a;
    return a + b;
}

// geometry of one time-independent Dense chain for the fragment engine (as chain_geo in rnde.hip)
static bool sde_geo(int n_layers, const int32_t* dims, const int32_t* act, ChainGeo& G) {
    G = ChainGeo{};
    if (n_layers < 1 || n_layers > kCMaxL) return false;
    G.n_layers = n_layers; G.time_dep = 0; G.pre_act = 0;
    int po = 0, fo = 0, bo = 0, to = 0;
    for (int l = 0; l <= n_layers; ++l) {
        if (dims[l] < 1 || dims[l] > 4 * kCMaxKs) return false;
        G.width[l] = dims[l]; G.nks[l] = (dims[l] + 3) / 4;
    }
    for (int l = 0; l < n_layers; ++l) {
        G.act[l] = act[l];
        G.poff[l] = po; po += G.width[l] * G.width[l + 1] + G.width[l + 1];
        G.foff[l] = fo; fo += ((G.nks[l + 1] + 3) / 4) * G.nks[l];
        G.boff[l] = bo; bo += 4 * ((G.nks[l + 1] + 3) / 4);
        G.toff[l] = to; to += ((G.nks[l] + 3) / 4) * G.nks[l + 1];
    }
    G.nfrag_f = fo; G.nfrag_b = bo; G.nfrag_t = to; G.nksD = G.nks[0];
    return true;
}

static void sde_tableau(int id, SriTableau& T, float& delta_default) {
    auto tri = [](float* M, double a21, double a31, double a32, double a41, double a42, double a43) {
        for (int i = 0; i < 16; ++i) M[i] = 0.f;
        M[4] = (float)a21; M[8] = (float)a31; M[9] = (float)a32; M[12] = (float)a41; M[13] = (float)a42; M[14] = (float)a43;
    };
    auto vec = [](float* v, double a, double b, double c, double d) { v[0] = (float)a; v[1] = (float)b; v[2] = (float)c; v[3] = (float)d; };
    delta_default = 1.f;
    if (id == RNDE_SDE_SRIW1) {   // Roessler 2010
        tri(T.A0, 0.75, 0, 0, 0, 0, 0); tri(T.A1, 0.25, 1, 0, 0, 0, 0.25); tri(T.B0, 1.5, 0, 0, 0, 0, 0); tri(T.B1, 0.5, -1, 0, -5, 3, 0.5);
        vec(T.alpha, 1.0 / 3, 2.0 / 3, 0, 0); vec(T.beta1, -1, 4.0 / 3, 2.0 / 3, 0); vec(T.beta2, -1, 4.0 / 3, -1.0 / 3, 0);
        vec(T.beta3, 2, -4.0 / 3, -2.0 / 3, 0); vec(T.beta4, -2, 5.0 / 3, -2.0 / 3, 1);
        delta_default = 1.f / 6.f;
    } else if (id == RNDE_SDE_SOSRI2) {   // Rackauckas & Nie 2018, stability-optimised for non-stiff... (SOSRI2)
        tri(T.A0, 0.13804532298278663, 0.5818361298250374, 0.4181638701749618, 0.4670018408674211, 0.8046204792187386, -0.27162232008616016);
        tri(T.A1, 0.45605532163856893, 0.7555807846451692, 0.24441921535482677, 0.6981181143266059, 0.3453277086024727, -0.04344582292908241);
        tri(T.B0, 0.08852381537667678, 1.0317752458971061, 0.4563552922077882, 1.73078280444124, -0.46089678470929774, -0.9637509618944188);
        tri(T.B1, 0.6753186815412179, -0.07452812525785148, -0.49783736486149366, -0.5591906709928903, 0.022696571806569924, -0.8984927888368557);
        vec(T.alpha, -0.15036858140642623, 0.7545275856696072, 0.686995463807979, -0.2911544680711602);
        vec(T.beta1, -0.45315689727309133, 0.8330937231303951, 0.3792843195533544, 0.24077885458934192);
        vec(T.beta2, -0.4994383733810986, 0.9181786186154077, -0.25613778661003145, -0.16260245862427797);
        vec(T.beta3, 1.4531568972730915, -0.8330937231303933, -0.3792843195533583, -0.24077885458934023);
        vec(T.beta4, -0.4976090683622265, 0.9148155835648892, -1.4102107084476505, 0.9930042001464879);
    } else {   // SOSRI (reference experiments/mnist_nsde.jl:49,:63)
        tri(T.A0, -0.04199224421316468, 2.842612915017106, -2.0527723684000727, 4.338237071435815, -2.8895936137439793, 2.3017575594644466);
        tri(T.A1, 0.26204282091330466, 0.20903646383505375, -0.1502377115150361, 0.05836595312746999, 0.6149440396332373, 0.08535117634046772);
        tri(T.B0, -0.21641093549612528, 1.5336352863679572, 0.26066223492647056, -1.0536037558179159, 1.7015284721089472, -0.20725685784180017);
        tri(T.B1, -0.5119011827621657, 2.67767339866713, -4.9395031322250995, 0.15580956238299215, 3.2361551006624674, -1.4223118283355949);
        vec(T.alpha, 1.140099274172029, -0.6401334255743456, 0.4736296532772559, 0.026404498125060714);
        vec(T.beta1, -1.8453464565104432, 2.688764531100726, -0.2523866501071323, 0.40896857551684956);
        vec(T.beta2, 0.4969658141589478, -0.5771202869753592, -0.12919702470322217, 0.2093514975196336);
        vec(T.beta3, 2.8453464565104425, -2.688764531100725, 0.2523866501071322, -0.40896857551684945);
        vec(T.beta4, 0.11522663875443433, -0.57877086147738, 0.2857851028163886, 0.17775911990655704);
    }
}

extern "C" rnde_status rnde_nsde_create(const rnde_nsde_config* c, rnde_nsde** out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c->device) { g_nsde_create_err = "no HIP device"; return RNDE_ERR_NO_DEVICE; }
    ChainGeo Gf, Gg;
    if (!sde_geo(c->drift_layers, c->drift_dims, c->drift_act, Gf) || !sde_geo(c->diff_layers, c->diff_dims, c->diff_act, Gg)) {
        g_nsde_create_err = "unsupported networks: Dense chains of 1..8 layers, every width 1..64"; return RNDE_ERR_BAD_ARG;
    }
    const int D = c->drift_dims[0];
    if (c->drift_dims[c->drift_layers] != D || c->diff_dims[0] != D || c->diff_dims[c->diff_layers] != D) {
        g_nsde_create_err = "drift and diffusion must map D -> D (diagonal noise)"; return RNDE_ERR_BAD_ARG;
    }
    if (c->solver < RNDE_SDE_SOSRI || c->solver > RNDE_SDE_SOSRI2 ||
        (c->regularize != RNDE_REG_NONE && c->regularize != RNDE_REG_ERR && c->regularize != RNDE_REG_STIFF) ||
        c->max_batch < 1 || c->max_attempts < 1 || c->max_attempts > 4000) {
        g_nsde_create_err = "solver / regularize / max_batch / max_attempts (<= 4000) out of range"; return RNDE_ERR_BAD_ARG;
    }
    if (c->regularize == RNDE_REG_STIFF && c->solver != RNDE_SDE_SOSRI2) {
        // StochasticDiffEq fills integrator.eigen_est only under a composite algorithm whose first method is SOSRI2 (AutoSOSRI2(SOSRI2()),
        // experiments/mnist_nsde.jl:60): its last two drift stages share one time, which is what makes the quotient an estimate of |J|
        g_nsde_create_err = "RNDE_REG_STIFF (the stiffness estimate of the SRI step) is defined for RNDE_SDE_SOSRI2 only"; return RNDE_ERR_BAD_ARG;
    }
    if (!(c->stability_size >= 0.f)) { g_nsde_create_err = "stability_size must be >= 0 (0 = 10.6)"; return RNDE_ERR_BAD_ARG; }
    rnde_nsde* h = new rnde_nsde();
    h->cfg = *c; h->D = D; h->Gf = Gf; h->Gg = Gg;
    h->Pf = chain_params(c->drift_layers, c->drift_dims); h->Pg = chain_params(c->diff_layers, c->diff_dims); h->P = h->Pf + h->Pg;
    h->NKD = D <= 16 ? 4 : (D <= 32 ? 8 : 16);
    h->fix = (c->drift_layers == 2 && c->diff_layers == 1 && Gf.nks[0] == 8 && Gf.nks[1] == 16 && c->generic == 0) ? 1 : 0;
    { const char* e = getenv("RNDE_SDE_MW"); h->mw = (h->fix && !(e && e[0] == '0')) ? 1 : 0; }
    float ddef = 1.f;
    sde_tableau(c->solver, h->T, ddef);
    h->order = 1.5f;
    h->beta2 = c->beta2 != 0.f ? c->beta2 : (float)(2.0 / (5.0 * 1.5));
    h->beta1 = c->beta1 != 0.f ? c->beta1 : (float)(7.0 / (10.0 * 1.5));
    h->gamma = c->gamma != 0.f ? c->gamma : 0.9f;
    h->qmin = c->qmin != 0.f ? c->qmin : 0.2f;
    h->qmax = c->qmax != 0.f ? c->qmax : 1.125f;
    h->qoldinit = c->qoldinit != 0.f ? c->qoldinit : 1e-4f;
    h->delta = c->delta != 0.f ? c->delta : ddef;
    h->Bpad_max = ((c->max_batch + 15) / 16) * 16;
    h->ntiles_max = h->Bpad_max / 16;
    h->nwg_max = (h->ntiles_max + kCW - 1) / kCW;
    if (h->nwg_max > 256) { g_nsde_create_err = "max_batch: at most 16384 columns (every workgroup of the one-launch solve must be resident)"; delete h; return RNDE_ERR_BAD_ARG; }
    const int cap = 2 * c->max_attempts + 8;
    const size_t uf = (size_t)((Gf.nfrag_f + Gf.nfrag_b + 3) / 4), ug = (size_t)((Gg.nfrag_f + Gg.nfrag_b + 3) / 4);
    h->lds_fwd = (uf + ug) * 1024 + (size_t)(56 + kSdeMaxOps * 8 + 5 * cap) * 4 + 64;
    h->lds_mw = (size_t)(kSmwLdsFloats + 56 + kSdeMaxOps * 8 + 5 * cap) * 4 + 64;
    h->xch_wg = h->mw ? std::max(h->nwg_max, std::min(h->ntiles_max, kSmwMaxTiles)) : h->nwg_max;
    const size_t ufb = (size_t)((Gf.nfrag_f + Gf.nfrag_b + Gf.nfrag_t + 3) / 4), ugb = (size_t)((Gg.nfrag_f + Gg.nfrag_b + Gg.nfrag_t + 3) / 4);
    h->lds_bwd = (ufb + ugb) * 1024 + 64;
    if (h->lds_fwd > 160 * 1024 || h->lds_bwd > 160 * 1024) { g_nsde_create_err = "networks too large: the weight fragments of both chains must fit the 160 KB LDS of a CU"; delete h; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { g_nsde_create_err = "hipSetDevice failed"; delete h; return RNDE_ERR_HIP; }
    const size_t A = (size_t)h->ntiles_max * h->NKD * 64;
    h->n_slots = cap;
    auto dm = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    bool ok = true;
    ok &= dm((void**)&h->frags_f, (size_t)(Gf.nfrag_f + Gf.nfrag_b + Gf.nfrag_t + 4) * 256);
    ok &= dm((void**)&h->frags_g, (size_t)(Gg.nfrag_f + Gg.nfrag_b + Gg.nfrag_t + 4) * 256);
    ok &= dm((void**)&h->slots, (size_t)h->n_slots * 2 * A * 4);
    ok &= dm((void**)&h->meta, (size_t)(c->max_attempts + 1) * sizeof(SdeMeta)) && dm((void**)&h->acc_meta, (size_t)(c->max_attempts + 1) * sizeof(SdeMeta));
    ok &= dm((void**)&h->fin, sizeof(SdeFinal)) && dm((void**)&h->abort_word, 16);
    ok &= dm((void**)&h->xch, (size_t)(c->max_attempts + 4) * 2 * h->xch_wg * 8);
    ok &= dm((void**)&h->svb, (size_t)(c->max_attempts + 1) * 4) && dm((void**)&h->replay, (size_t)(c->max_attempts + 1) * 8);
    ok &= dm((void**)&h->part, (size_t)h->nwg_max * 4);
    if (c->regularize == RNDE_REG_STIFF) ok &= dm((void**)&h->eigpart, (size_t)c->max_attempts * 2 * h->xch_wg * 4);      // per-workgroup partials of the estimate's two norms
    ok &= hipHostMalloc((void**)&h->h_meta, (size_t)(c->max_attempts + 1) * sizeof(SdeMeta)) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_acc_meta, (size_t)(c->max_attempts + 1) * sizeof(SdeMeta)) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_fin, sizeof(SdeFinal)) == hipSuccess;
    ok &= dm((void**)&h->xcc, 256 * 4) && hipHostMalloc((void**)&h->h_xcc, 256 * 4) == hipSuccess;
    { const char* e = getenv("RNDE_SDE_LOCAL"); if (e && e[0] == '0') h->xch_local = 0; }
    ok &= hipHostMalloc((void**)&h->h_svb, (size_t)(c->max_attempts + 1) * 4) == hipSuccess;
    ok &= hipHostMalloc((void**)&h->h_part, (size_t)h->nwg_max * 4) == hipSuccess;
    if (!ok) { g_nsde_create_err = "device allocation failed"; rnde_nsde_destroy(h); return RNDE_ERR_HIP; }
    for (auto& e : h->tev) if (hipEventCreate(&e) != hipSuccess) { g_nsde_create_err = "hipEventCreate failed"; rnde_nsde_destroy(h); return RNDE_ERR_HIP; }
    hipMemset(h->abort_word, 0, 16);
    hipMemset(h->xch, 0, (size_t)(c->max_attempts + 4) * 2 * h->xch_wg * 8);
    hipMemset(h->frags_f, 0, (size_t)(Gf.nfrag_f + Gf.nfrag_b + Gf.nfrag_t + 4) * 256);
    hipMemset(h->frags_g, 0, (size_t)(Gg.nfrag_f + Gg.nfrag_b + Gg.nfrag_t + 4) * 256);
    *out = h;
    return RNDE_OK;
}

extern "C" void rnde_nsde_destroy(rnde_nsde* h) {
    if (!h) return;
    void* d[] = {h->eigpart, h->frags_f, h->frags_g, h->slots, h->tape, h->noise, h->replay, h->meta, h->acc_meta, h->fin, h->xch, h->abort_word, h->svb,
                 h->slab_f, h->slab_g, h->wslab, h->wslab_r, h->ev_t, h->part, h->sv_t_dev, h->head_ws, h->cg_ws, h->xcc};
    for (void* p : d) if (p) (void)hipFree(p);
    void* hd[] = {h->h_meta, h->h_acc_meta, h->h_fin, h->h_svb, h->h_part, h->h_xcc};
    for (void* p : hd) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : h->tev) if (e) (void)hipEventDestroy(e);
    if (h->ev_host) (void)hipEventDestroy(h->ev_host);
    delete h;
}

static SdeParams sde_params(rnde_nsde* h, const float* x, const float* noise, int n_pool, int B, float t0, float t1, int keep_tape) {
    SdeParams Q{};
    Q.Gf = h->Gf; Q.Gg = h->Gg; Q.frags_f = h->frags_f; Q.frags_g = h->frags_g; Q.T = h->T;
    Q.x = x; Q.noise = noise; Q.slots = h->slots; Q.tape = h->tape; Q.meta = h->meta; Q.fin = h->fin; Q.xch = h->xch; Q.abort_word = h->abort_word;
    Q.u_out = nullptr; Q.replay = nullptr; Q.n_replay = 0;
    Q.D = h->D; Q.B = B; Q.ntiles = (B + 15) / 16; Q.nwg = (Q.ntiles + kCW - 1) / kCW;
    Q.n_pool = n_pool; Q.n_slots = h->n_slots; Q.max_attempts = h->cfg.max_attempts; Q.keep_tape = keep_tape; Q.reg_kind = h->cfg.regularize;
    Q.epoch = h->epoch; Q.t0 = t0; Q.t1 = t1; Q.reltol = h->cfg.reltol; Q.abstol = h->cfg.abstol;
    Q.beta1 = h->beta1; Q.beta2 = h->beta2; Q.gamma = h->gamma; Q.qmin = h->qmin; Q.qmax = h->qmax; Q.qoldinit = h->qoldinit; Q.delta = h->delta;
    Q.order = h->order;
    Q.eigpart = h->eigpart; Q.stab = h->cfg.stability_size > 0.f ? h->cfg.stability_size : 10.6f;      // StochasticDiffEq.alg_stability_size(SOSRI2())
    return Q;
}

static rnde_status sde_pack(rnde_nsde* h, const float* p_dev, hipStream_t s) {
    const long long tf = (long long)(h->Gf.nfrag_f + h->Gf.nfrag_b + h->Gf.nfrag_t) * 64, tg = (long long)(h->Gg.nfrag_f + h->Gg.nfrag_b + h->Gg.nfrag_t) * 64;
    hipLaunchKernelGGL(rnde_chain_pack_kernel, dim3((int)std::min<long long>((tf + 255) / 256, 512)), dim3(256), 0, s, p_dev, h->frags_f, h->Gf);
    hipLaunchKernelGGL(rnde_chain_pack_kernel, dim3((int)std::min<long long>((tg + 255) / 256, 512)), dim3(256), 0, s, p_dev + h->Pf, h->frags_g, h->Gg);
    SCHK(h, hipGetLastError());
    return RNDE_OK;
}

template <int NKD, int FIXH = 0>
static hipError_t launch_solve(rnde_nsde* h, const SdeParams& Q, hipStream_t s) {
    static DeviceOnce attr;
    if (attr.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_sde_solve_kernel<NKD, FIXH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr.done();
    }
    hipLaunchKernelGGL((rnde_sde_solve_kernel<NKD, FIXH>), dim3(Q.nwg), dim3(64 * kCW), h->lds_fwd, s, Q);
    return hipGetLastError();
}
template <int NKD>
static hipError_t launch_attempt(rnde_nsde* h, const SdeParams& Q, const float* up, const float* dW, const float* dZ, float dt, float* kg, float* un, hipStream_t s) {
    static DeviceOnce attr;
    if (attr.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_sde_attempt_kernel<NKD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr.done();
    }
    hipLaunchKernelGGL(rnde_sde_attempt_kernel<NKD>, dim3(Q.nwg), dim3(64 * kCW), h->lds_fwd, s, Q, up, dW, dZ, dt, kg, un, h->part);
    return hipGetLastError();
}
template <int NKD, int FIXH = 0>
static hipError_t launch_bwd(rnde_nsde* h, const SdeBwdParams& Bq, hipStream_t s) {
    static DeviceOnce attr;
    if (attr.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_sde_bwd_kernel<NKD, FIXH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr.done();
    }
    hipLaunchKernelGGL((rnde_sde_bwd_kernel<NKD, FIXH>), dim3(Bq.F.nwg), dim3(64 * kCW), h->lds_bwd, s, Bq);
    return hipGetLastError();
}

static rnde_status nsde_forward_impl(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, const float* noise_dev,
                                     int32_t n_pool, uint64_t seed, const float* steps_host, int32_t n_steps, float* u_out_dev, int64_t* nfe1_out,
                                     int64_t* nfe2_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream,
                                     const float* saveat_host = nullptr, int32_t n_saveat = 0, float* sv_out_dev = nullptr) {
    if (!h) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_saveat > 0) {
        for (int i = 0; i < n_saveat; ++i)
            if (!(saveat_host[i] >= t0 && saveat_host[i] <= t1) || (i > 0 && !(saveat_host[i] > saveat_host[i - 1]))) {
                h->err = "saveat must be increasing and inside [t0, t1]"; return RNDE_ERR_BAD_ARG;
            }
        if ((size_t)n_saveat > h->sv_cap) {
            if (h->sv_t_dev) (void)hipFree(h->sv_t_dev);
            h->sv_t_dev = nullptr; h->sv_cap = 0;
            SCHK(h, hipMalloc((void**)&h->sv_t_dev, (size_t)n_saveat * 4));
            h->sv_cap = n_saveat;
        }
        h->saveat.assign(saveat_host, saveat_host + n_saveat);
        SCHK(h, hipMemcpyAsync(h->sv_t_dev, h->saveat.data(), (size_t)n_saveat * 4, hipMemcpyHostToDevice, s));
    } else h->saveat.clear();
    if (B < 1 || B > h->cfg.max_batch || !(t1 > t0) || !x_dev || !p_dev) { h->err = "bad B, tspan or pointers"; return RNDE_ERR_BAD_ARG; }
    if (noise_dev && n_pool < 1) { h->err = "noise pool: n_pool >= 1"; return RNDE_ERR_BAD_ARG; }
    if (n_steps > h->cfg.max_attempts) { h->err = "replay: more steps than max_attempts"; return RNDE_ERR_BAD_ARG; }
    SCHK(h, hipSetDevice(h->cfg.device));
    h->have_tape = false;
    const int ntiles = (B + 15) / 16;
    const size_t A = (size_t)ntiles * h->NKD * 64;
    if (keep_tape) {
        const size_t need = (size_t)h->cfg.max_attempts * 12 * A;
        if (h->tape_floats < need) {
            if (h->tape) (void)hipFree(h->tape);
            h->tape = nullptr; h->tape_floats = 0;
            SCHK(h, hipMalloc((void**)&h->tape, need * 4));
            h->tape_floats = need;
        }
    }
    bool lib_noise = false;
    if (!noise_dev) {   // the library's own stream: one pool per solve from (seed, epoch-independent: the seed alone names the path)
        // Draw k of the stream depends on (seed, k) alone (counter-based generator), so the pool may be any prefix: it is filled for one and a half
        // times the draws the last solve consumed (+ 16) instead of for max_attempts + 1 (34 MB, 17 us in front of every solve at B = 512), and a
        // solve that runs out of it (status 4) is redone with the full pool -- same draws, same result.
        lib_noise = true;
        n_pool = (h->pool_pred > 0 && n_steps == 0) ? std::min(h->pool_pred, h->cfg.max_attempts + 1) : h->cfg.max_attempts + 1;
        const size_t need = (size_t)n_pool * 2 * h->D * B;
        if (h->noise_floats < need) {
            if (h->noise) (void)hipFree(h->noise);
            h->noise = nullptr; h->noise_floats = 0;
            SCHK(h, hipMalloc((void**)&h->noise, need * 4));
            h->noise_floats = need;
        }
        hipLaunchKernelGGL(rnde_normal_fill_kernel, dim3((unsigned)std::min<size_t>((need / 4 + 255) / 256, 4096)), dim3(256), 0, s, h->noise, (long long)need, (unsigned long long)seed, 0ull);
        SCHK(h, hipGetLastError());
        noise_dev = h->noise;
    }
    rnde_status st = sde_pack(h, p_dev, s);
    if (st != RNDE_OK) return st;
    ++h->epoch;
    if (h->epoch >= 500000u) {   // tags are epoch * 8192 + sequence: start over (entries are rewritten before they are read)
        h->epoch = 1;
        SCHK(h, hipMemsetAsync(h->xch, 0, (size_t)(h->cfg.max_attempts + 4) * 2 * h->xch_wg * 8, s));
    }
    SdeParams Q = sde_params(h, x_dev, noise_dev, n_pool, B, t0, t1, keep_tape ? 1 : 0);
    Q.u_out = u_out_dev;
    Q.sv_t = n_saveat > 0 ? h->sv_t_dev : nullptr; Q.nsave = n_saveat; Q.sv_out = sv_out_dev;
    if (n_steps > 0) {
        SCHK(h, hipMemcpyAsync(h->replay, steps_host, (size_t)n_steps * 8, hipMemcpyHostToDevice, s));
        Q.replay = h->replay; Q.n_replay = n_steps;
    }
    h->tev_f = false;
    SCHK(h, hipEventRecord(h->tev[0], s));
    hipError_t e;
    bool local_xch = false;
    // one workgroup of four waves per tile, all of them resident (they meet once per attempt): 144 VGPRs and ~20 KB of LDS let a CU hold two,
    // so the limit is 512 tiles = 8,192 columns (the reference's evaluation call with trajectories = 10 is 5,120: mnist_nsde.jl:154-155)
    if (h->mw && ntiles <= kSmwMaxTiles) {
        static DeviceOnce attr;
        if (attr.need()) { SCHK(h, hipFuncSetAttribute((const void*)rnde_sde_solve_mw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr.done(); }
        Q.nwg = ntiles;
        local_xch = h->xch_local && ntiles <= 32;      // one XCD has 32 CUs: one workgroup each
        Q.xch_local = local_xch ? 1 : 0; Q.xcc = h->xcc;
        hipLaunchKernelGGL(rnde_sde_solve_mw_kernel, dim3(local_xch ? 8 * ntiles : ntiles), dim3(kSmwThreads), h->lds_mw, s, Q);
        e = hipGetLastError();
        if (local_xch && e == hipSuccess) e = hipMemcpyAsync(h->h_xcc, h->xcc, (size_t)ntiles * 4, hipMemcpyDeviceToHost, s);
    } else
        e = h->fix ? launch_solve<8, 16>(h, Q, s)
                   : (h->NKD == 4 ? launch_solve<4>(h, Q, s) : (h->NKD == 8 ? launch_solve<8>(h, Q, s) : launch_solve<16>(h, Q, s)));
    SCHK(h, e);
    if (h->cfg.regularize == RNDE_REG_STIFF) {      // the two norms of every attempt's stiffness estimate -> meta[n].n1 / .n2 (fixed-order sums of the workgroups' partials)
        hipLaunchKernelGGL(rnde_sde_eig_reduce_kernel, dim3(h->cfg.max_attempts), dim3(64), 0, s, Q);
        SCHK(h, hipGetLastError());
    }
    SCHK(h, hipEventRecord(h->tev[1], s));
    h->tev_f = true;
    SCHK(h, hipMemcpyAsync(h->h_fin, h->fin, sizeof(SdeFinal), hipMemcpyDeviceToHost, s));
    SCHK(h, hipMemcpyAsync(h->h_meta, h->meta, (size_t)h->cfg.max_attempts * sizeof(SdeMeta), hipMemcpyDeviceToHost, s));
    if (h->after_solve) {   // (fused training step) wait for the copies only; what the hook queues runs while the host wakes up
        SCHK(h, hipEventRecord(h->ev_host, s));
        const rnde_status hs = h->after_solve(s);
        if (hs != RNDE_OK) return hs;
        SCHK(h, hipEventSynchronize(h->ev_host));
    } else SCHK(h, hipStreamSynchronize(s));
    if (local_xch) {   // did the workgroups really share an XCD?  If not, their meeting had no coherent meeting place: redo the solve the safe way, for good
        bool same = true;
        for (int i = 1; i < ntiles; ++i) same = same && h->h_xcc[i] == h->h_xcc[0];
        if (!same) {
            fprintf(stderr, "[rnde] SDE solve: workgroups pinned by block index landed on different XCDs; using the placement-independent exchange from now on\n");
            h->xch_local = 0;
            return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, lib_noise ? nullptr : noise_dev, lib_noise ? 0 : n_pool, seed, steps_host, n_steps, u_out_dev, nfe1_out, nfe2_out,
                                     saveval_host, n_saveval_out, keep_tape, stream, saveat_host, n_saveat, sv_out_dev);
        }
    }
    const SdeFinal F = *h->h_fin;
    h->B = B; h->ntiles = ntiles; h->nwg = Q.nwg; h->n_att = F.n_att; h->n_acc = F.n_acc; h->n_draws = F.n_draws; h->t0 = t0;
    if (nfe1_out) *nfe1_out = 2 + 4 * (int64_t)F.n_att;   // the closures' counters (neural_sde.jl:46,:50): 2 probes of the initial-step rule + 4 per attempt
    if (nfe2_out) *nfe2_out = 2 + 4 * (int64_t)F.n_att;
    int nsv = 0;
    h->sv_index.clear();
    if (h->cfg.regularize != RNDE_REG_NONE) {
        // the experiment's save_func on the integrator after every accepted step (mnist_nsde.jl:48: EEst * dt; :53-58: |eigen_est| / stability_size, zero
        // and NaN estimates recorded as 0), and once at the callback's initialisation (EEst = 1, dt = 0, eigen_est = 1)
        const bool stiff = h->cfg.regularize == RNDE_REG_STIFF;
        auto value = [&](float eest, float dt, float eigen) {
            if (!stiff) return eest * dt;
            const float a = std::fabs(eigen);
            return (a == 0.f || a != a) ? 0.f : a / Q.stab;
        };
        if (h->cfg.cb_save_start) { if (saveval_host) saveval_host[nsv] = value(1.f, 0.f, 1.f); ++nsv; }
        for (int i = 0; i < F.n_att; ++i)
            if (h->h_meta[i].accepted) {
                const SdeMeta& m = h->h_meta[i];
                if (saveval_host) saveval_host[nsv] = value(m.eest, m.dt, stiff ? m.n1 / m.n2 : 0.f);
                h->sv_index.push_back(nsv++);
            }
    }
    h->n_saveval = nsv;
    if (n_saveval_out) *n_saveval_out = nsv;
    switch (F.status) {
        case 0: break;
        case 1: h->err = "max_attempts reached"; return RNDE_ERR_MAX_ATTEMPTS;
        case 2: h->err = "dt underflow"; return RNDE_ERR_DT_UNDERFLOW;
        case 3: h->err = "non-finite error estimate or dt"; return RNDE_ERR_NONFINITE;
        case 4:
            if (lib_noise && n_pool < h->cfg.max_attempts + 1) {   // the shortened pool of the library's own stream ran out: the whole pool, once more
                h->pool_pred = 0;
                return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, nullptr, 0, seed, steps_host, n_steps, u_out_dev, nfe1_out, nfe2_out, saveval_host, n_saveval_out,
                                         keep_tape, stream, saveat_host, n_saveat, sv_out_dev);
            }
            h->err = "noise pool or stack capacity exhausted (n_pool must cover 1 + attempts draws)"; return RNDE_ERR_BAD_ARG;
        default:
            (void)hipMemsetAsync(h->abort_word, 0, 16, s);
            h->err = "a workgroup of the one-launch solve timed out waiting for the others (not all resident?)";
            return RNDE_ERR_HIP;
    }
    if (lib_noise) h->pool_pred = std::min(h->cfg.max_attempts + 1, F.n_draws + F.n_draws / 2 + 16);
    h->have_tape = keep_tape != 0;
    return RNDE_OK;
}

extern "C" rnde_status rnde_nsde_forward(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, const float* noise_dev,
                                         int32_t n_pool, uint64_t seed, float* u_out_dev, int64_t* nfe1_out, int64_t* nfe2_out, float* saveval_host,
                                         int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, noise_dev, n_pool, seed, nullptr, 0, u_out_dev, nfe1_out, nfe2_out, saveval_host, n_saveval_out, keep_tape, stream);
}
extern "C" rnde_status rnde_nsde_forward_saveat(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, const float* noise_dev,
                                                int32_t n_pool, uint64_t seed, const float* saveat_host, int32_t n_saveat, float* u_saved_dev,
                                                int64_t* nfe1_out, int64_t* nfe2_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape,
                                                void* stream) {
    if (!h || !saveat_host || n_saveat < 1 || !u_saved_dev) return RNDE_ERR_BAD_ARG;
    return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, noise_dev, n_pool, seed, nullptr, 0, nullptr, nfe1_out, nfe2_out, saveval_host, n_saveval_out, keep_tape,
                             stream, saveat_host, n_saveat, u_saved_dev);
}
// save_everystep = true of the SDE layer (reference src/models/neural_sde.jl:14): the state after every accepted step (t0 first when save_start).
// As rnde_node_forward_everystep: the solve runs twice on the SAME noise (the explicit pool, or the library's generator with the same seed: the draws
// are keyed by their index), the second time saving at the first's accepted step ends -- the value saved at a step's end is u_new itself.
extern "C" rnde_status rnde_nsde_forward_everystep(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, const float* noise_dev,
                                                   int32_t n_pool, uint64_t seed, int32_t save_start, float* sol_out_dev, int32_t capacity, float* t_host_out,
                                                   int32_t* n_out, int64_t* nfe1_out, int64_t* nfe2_out, float* saveval_host, int32_t* n_saveval_out,
                                                   int32_t keep_tape, void* stream) {
    if (!h || !n_out || !sol_out_dev || capacity < 1) return RNDE_ERR_BAD_ARG;
    rnde_status st = nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, noise_dev, n_pool, seed, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
    if (st != RNDE_OK) return st;
    std::vector<float> times;
    if (save_start) times.push_back(t0);
    for (int i = 0; i < h->n_att; ++i)
        if (h->h_meta[i].accepted) { float tn = h->h_meta[i].t + h->h_meta[i].dt; if (tn > t1) tn = t1; times.push_back(tn); }
    *n_out = (int32_t)times.size();
    if ((int)times.size() > capacity) { h->err = "rnde_nsde_forward_everystep: more accepted steps than the output has room for"; return RNDE_ERR_BAD_ARG; }
    if (t_host_out) memcpy(t_host_out, times.data(), times.size() * sizeof(float));
    return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, noise_dev, n_pool, seed, nullptr, 0, nullptr, nfe1_out, nfe2_out, saveval_host, n_saveval_out, keep_tape,
                             stream, times.data(), (int32_t)times.size(), sol_out_dev);
}
extern "C" rnde_status rnde_nsde_forward_replay(rnde_nsde* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, const float* noise_dev,
                                                int32_t n_pool, const float* steps_host, int32_t n_steps, float* u_out_dev, int64_t* nfe1_out,
                                                int64_t* nfe2_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    if (!h || !steps_host || n_steps < 1 || !noise_dev) return RNDE_ERR_BAD_ARG;
    return nsde_forward_impl(h, x_dev, p_dev, B, t0, t1, noise_dev, n_pool, 0, steps_host, n_steps, u_out_dev, nfe1_out, nfe2_out, saveval_host, n_saveval_out, keep_tape, stream);
}

extern "C" rnde_status rnde_nsde_steps(rnde_nsde* h, float* steps_host, int32_t capacity, int32_t* n_out, int32_t* draws_out) {
    if (!h) return RNDE_ERR_BAD_ARG;
    const int n = std::min(capacity, h->n_att);
    for (int i = 0; i < n; ++i) {
        steps_host[4 * i] = h->h_meta[i].t; steps_host[4 * i + 1] = h->h_meta[i].dt; steps_host[4 * i + 2] = h->h_meta[i].eest;
        steps_host[4 * i + 3] = h->h_meta[i].accepted ? 1.f : 0.f;
    }
    if (n_out) *n_out = h->n_att;
    if (draws_out) *draws_out = h->n_draws;
    return RNDE_OK;
}

extern "C" rnde_status rnde_nsde_debug_attempt(rnde_nsde* h, const float* uprev_dev, const float* p_dev, int32_t B, float dt, const float* dW_dev,
                                               const float* dZ_dev, float* kg_out_dev, float* unew_out_dev, float* eest_out, void* stream) {
    if (!h || B < 1 || B > h->cfg.max_batch) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    SCHK(h, hipSetDevice(h->cfg.device));
    h->have_tape = false;
    rnde_status st = sde_pack(h, p_dev, s);
    if (st != RNDE_OK) return st;
    SdeParams Q = sde_params(h, uprev_dev, nullptr, 0, B, 0.f, 1.f, 0);
    hipError_t e = h->NKD == 4 ? launch_attempt<4>(h, Q, uprev_dev, dW_dev, dZ_dev, dt, kg_out_dev, unew_out_dev, s)
                 : (h->NKD == 8 ? launch_attempt<8>(h, Q, uprev_dev, dW_dev, dZ_dev, dt, kg_out_dev, unew_out_dev, s)
                                : launch_attempt<16>(h, Q, uprev_dev, dW_dev, dZ_dev, dt, kg_out_dev, unew_out_dev, s));
    SCHK(h, e);
    SCHK(h, hipMemcpyAsync(h->h_part, h->part, (size_t)Q.nwg * 4, hipMemcpyDeviceToHost, s));
    SCHK(h, hipStreamSynchronize(s));
    double ss = 0;
    for (int i = 0; i < Q.nwg; ++i) ss += (double)h->h_part[i];
    if (eest_out) *eest_out = (float)std::sqrt(ss / ((double)h->D * (double)B));
    return RNDE_OK;
}

static rnde_status nsde_backward_impl(rnde_nsde* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev, float* p_bar_dev,
                                      void* stream, bool sync) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (!h->have_tape) { h->err = "no recorded forward"; return RNDE_ERR_NO_TAPE; }
    hipStream_t s = (hipStream_t)stream;
    SCHK(h, hipSetDevice(h->cfg.device));
    const int n_acc = h->n_acc, ntiles = h->ntiles;
    int a = 0;
    for (int i = 0; i < h->n_att; ++i)
        if (h->h_meta[i].accepted) {
            h->h_acc_meta[a] = h->h_meta[i];
            h->h_svb[a] = (saveval_bar_host && a < (int)h->sv_index.size()) ? saveval_bar_host[h->sv_index[a]] : 0.f;
            ++a;
        }
    if (a != n_acc) { h->err = "tape bookkeeping mismatch"; return RNDE_ERR_NO_TAPE; }
    SCHK(h, hipMemcpyAsync(h->acc_meta, h->h_acc_meta, (size_t)std::max(1, n_acc) * sizeof(SdeMeta), hipMemcpyHostToDevice, s));
    SCHK(h, hipMemcpyAsync(h->svb, h->h_svb, (size_t)std::max(1, n_acc) * 4, hipMemcpyHostToDevice, s));
    SdeBwdParams Bq{};
    Bq.F = sde_params(h, nullptr, nullptr, 0, h->B, 0.f, 1.f, 1);
    Bq.ubar = u_bar_dev; Bq.xbar = x_bar_dev; Bq.svb_acc = h->svb; Bq.acc_meta = h->acc_meta; Bq.n_acc = n_acc;
    Bq.nsave = (int)h->saveat.size(); Bq.sv_t = Bq.nsave ? h->sv_t_dev : nullptr; Bq.save_t0 = (Bq.nsave && h->saveat[0] == h->t0) ? 1 : 0;
    auto pad4 = [](int k) { return 4 * ((k + 3) / 4); };
    auto dump = [&](const ChainGeo& G, BChainParams& C) {
        C.G = G; C.ntiles = ntiles;
        int row = 0;
        for (int l = 0; l < G.n_layers; ++l) { C.hrow[l] = row; row += pad4(G.nks[l]); C.zrow[l] = row; row += pad4(G.nks[l + 1]); }
        C.hrow[G.n_layers] = row; row += pad4(G.nks[G.n_layers]);
        C.RS = row; C.ev_stride = (long long)ntiles * row * 64;
    };
    dump(h->Gf, Bq.Cf); dump(h->Gg, Bq.Cg);
    const int n_evals = 4 * std::max(1, n_acc);
    // Buffers that scale with the number of accepted steps grow by half as much again (and 16 steps) when they have to: while a
    // model trains, the step count creeps up by one or two per training step, and a hipFree + hipMalloc per step stalled the GPU for
    // ~0.4 ms of a 1.6 ms step (the free waits for the device).
    const size_t n_evals_cap = 4 * ((size_t)std::max(1, n_acc) * 3 / 2 + 16);
    auto ensure = [&](float*& p, size_t& have, size_t need, size_t grow_to) -> bool {
        if (have >= need) return true;
        if (p) (void)hipFree(p);
        p = nullptr; have = 0;
        if (hipMalloc((void**)&p, grow_to * 4) != hipSuccess) {
            if (hipMalloc((void**)&p, need * 4) != hipSuccess) return false;   // (no room for the head room: exactly what is needed)
            grow_to = need;
        }
        have = grow_to;
        return true;
    };
    if (!ensure(h->slab_f, h->slab_f_floats, (size_t)n_evals * Bq.Cf.ev_stride, n_evals_cap * Bq.Cf.ev_stride) ||
        !ensure(h->slab_g, h->slab_g_floats, (size_t)n_evals * Bq.Cg.ev_stride, n_evals_cap * Bq.Cg.ev_stride)) {
        h->err = "slab allocation failed"; return RNDE_ERR_HIP;
    }
    if (h->ev_t_n < (size_t)n_evals) {
        if (h->ev_t) (void)hipFree(h->ev_t);
        h->ev_t = nullptr; h->ev_t_n = 0;
        SCHK(h, hipMalloc((void**)&h->ev_t, n_evals_cap * 4));
        SCHK(h, hipMemsetAsync(h->ev_t, 0, n_evals_cap * 4, s));
        h->ev_t_n = n_evals_cap;
    }
    if (!h->wslab) { SCHK(h, hipMalloc((void**)&h->wslab, (size_t)96 * h->P * 4)); SCHK(h, hipMalloc((void**)&h->wslab_r, (size_t)16 * h->P * 4)); }
    Bq.Cf.slab = h->slab_f; Bq.Cg.slab = h->slab_g;
    if (n_acc == 0) {   // nothing was integrated: identity (never with saveat: t1 > t0 guarantees a step)
        if (Bq.nsave) { h->err = "reverse pass of a saveat solve without accepted steps"; return RNDE_ERR_NO_TAPE; }
        SCHK(h, hipMemcpyAsync(x_bar_dev, u_bar_dev, (size_t)h->D * h->B * 4, hipMemcpyDeviceToDevice, s));
        SCHK(h, hipMemsetAsync(p_bar_dev, 0, (size_t)h->P * 4, s));
        SCHK(h, hipStreamSynchronize(s));
        h->have_tape = false;
        return RNDE_OK;
    }
    h->tev_b = false;
    SCHK(h, hipEventRecord(h->tev[2], s));
    hipError_t e;
    if (h->mw) {   // four waves per tile (rnde_sdemw.h); no residency requirement here: the reverse sweep has no meeting
        hipLaunchKernelGGL(rnde_sde_bwd_mw_kernel, dim3(Bq.F.ntiles), dim3(kSmwThreads), (size_t)kSmwBwdLdsFloats * 4, s, Bq);
        e = hipGetLastError();
    } else
        e = h->fix ? launch_bwd<8, 16>(h, Bq, s)
                   : (h->NKD == 4 ? launch_bwd<4>(h, Bq, s) : (h->NKD == 8 ? launch_bwd<8>(h, Bq, s) : launch_bwd<16>(h, Bq, s)));
    SCHK(h, e);
    SCHK(h, hipEventRecord(h->tev[3], s));
    h->tev_b = true;
    // parameter gradients of both chains over all evaluations (rnde_bchain.h), per-chunk partials in Flux.destructure order
    const int n_units = 4 * n_acc * ntiles;
    const int chunks = std::max(1, std::min(96, n_units / 8));
    const int per_chunk = (n_units + chunks - 1) / chunks;
    hipLaunchKernelGGL(rnde_chain_wgrad_kernel, dim3(h->Gf.n_layers, chunks), dim3(64 * kCW), 0, s, Bq.Cf, (const float*)h->ev_t, n_units, per_chunk, h->wslab, h->P);
    hipLaunchKernelGGL(rnde_chain_wgrad_kernel, dim3(h->Gg.n_layers, chunks), dim3(64 * kCW), 0, s, Bq.Cg, (const float*)h->ev_t, n_units, per_chunk, h->wslab + h->Pf, h->P);
    SCHK(h, hipGetLastError());
    {
        const long long len = h->P;
        const int grid = (int)std::min<long long>((len + 255) / 256, 2048);
        if (chunks <= 16) hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)h->wslab, chunks, chunks, len, p_bar_dev);
        else {
            const int per_group = (chunks + 15) / 16, groups = (chunks + per_group - 1) / per_group;
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, groups), dim3(256), 0, s, (const float*)h->wslab, chunks, per_group, len, h->wslab_r);
            hipLaunchKernelGGL(rnde_wgrad_reduce, dim3(grid, 1), dim3(256), 0, s, (const float*)h->wslab_r, groups, groups, len, p_bar_dev);
        }
        SCHK(h, hipGetLastError());
    }
    if (sync) SCHK(h, hipStreamSynchronize(s));     // (nothing is read back: the synchronising form only makes "returned" mean "finished")
    h->have_tape = false;
    return RNDE_OK;
}
extern "C" rnde_status rnde_nsde_backward(rnde_nsde* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev, float* p_bar_dev,
                                          void* stream) {
    return nsde_backward_impl(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, stream, true);
}
extern "C" rnde_status rnde_nsde_backward_async(rnde_nsde* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev, float* p_bar_dev,
                                                void* stream) {
    return nsde_backward_impl(h, u_bar_dev, saveval_bar_host, x_bar_dev, p_bar_dev, stream, false);
}

// postsde Dense(D, C) + logitcrossentropy and their reverse for ClassifierNSDE with one trajectory per input (the kernels of rnde_head.h,
// as rnde_classifier_head for the ODE classifier; reference src/models/supervised_classification.jl:96-97 + experiments/mnist_nsde.jl loss)
static rnde_status nsde_head_reserve(rnde_nsde* h, int32_t B, int32_t n_classes) {
    const size_t need = (size_t)B * n_classes + B + (size_t)kHeadChunks * n_classes * h->D;
    if (h->head_ws_floats < need) {
        if (h->head_ws) (void)hipFree(h->head_ws);
        h->head_ws = nullptr; h->head_ws_floats = 0;
        SCHK(h, hipMalloc((void**)&h->head_ws, need * 4));
        h->head_ws_floats = need;
    }
    return RNDE_OK;
}
extern "C" rnde_status rnde_nsde_classifier_head(rnde_nsde* h, const float* u_dev, const float* p3_dev, const float* y_dev, int32_t B, int32_t n_classes,
                                                 float* logits_out_dev, float* u_bar_dev, float* p3_bar_dev, float* ce_out_dev, void* stream) {
    if (!h || B < 1 || n_classes < 1 || n_classes > kHeadMaxC) return RNDE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const rnde_status rs = nsde_head_reserve(h, B, n_classes);
    if (rs != RNDE_OK) return rs;
    float* delta = h->head_ws;
    float* ce_col = h->head_ws + (size_t)B * n_classes;
    float* partial = ce_col + B;
    if (n_classes == 10) hipLaunchKernelGGL((rnde_head_col_kernel<10>), dim3(B), dim3(256), 0, s, u_dev, p3_dev, y_dev, h->D, n_classes, B, logits_out_dev, u_bar_dev, delta, ce_col);
    else hipLaunchKernelGGL((rnde_head_col_kernel<0>), dim3(B), dim3(256), 0, s, u_dev, p3_dev, y_dev, h->D, n_classes, B, logits_out_dev, u_bar_dev, delta, ce_col);
    hipLaunchKernelGGL(rnde_head_wgrad_kernel, dim3((h->D + 255) / 256, kHeadChunks), dim3(256), 0, s, u_dev, (const float*)delta, h->D, n_classes, B, partial);
    hipLaunchKernelGGL(rnde_head_reduce_kernel, dim3((n_classes * h->D + 255) / 256), dim3(256), 0, s, (const float*)partial, (const float*)delta, (const float*)ce_col,
                       h->D, n_classes, B, p3_bar_dev, ce_out_dev);
    SCHK(h, hipGetLastError());
    return RNDE_OK;
}

// One training-step gradient of the ClassifierNSDE loss around the SDE solve (one trajectory per input) in ONE call: forward solve (taped),
// postsde Dense + logitcrossentropy and their reverse, reverse sweep -- rnde_nsde_forward + rnde_nsde_classifier_head + rnde_nsde_backward_async
// with the head queued before the forward's host wait (the three separate calls left the GPU idle for ~0.1 ms of a 1.3 ms step).
extern "C" rnde_status rnde_nsde_classifier_grad(rnde_nsde* h, const float* x_dev, const float* p2_dev, const float* p3_dev, const float* y_dev,
                                                 int32_t B, int32_t n_classes, float t0, float t1, const float* noise_dev, int32_t n_pool,
                                                 uint64_t seed, float lambda, float* p2_bar_dev, float* p3_bar_dev, float* x_bar_dev,
                                                 float* ce_out_dev, float* reg_out_host, int64_t* nfe1_out, int64_t* nfe2_out, void* stream) {
    if (!h || !x_dev || !p2_dev || !p3_dev || !y_dev || !p2_bar_dev || !p3_bar_dev || !x_bar_dev || !ce_out_dev) return RNDE_ERR_BAD_ARG;
    if (B < 1 || B > h->cfg.max_batch || n_classes < 1 || n_classes > kHeadMaxC) return RNDE_ERR_BAD_ARG;
    SCHK(h, hipSetDevice(h->cfg.device));
    const size_t A = (size_t)h->D * B;
    if (h->cg_ws_floats < 2 * A) {
        if (h->cg_ws) (void)hipFree(h->cg_ws);
        h->cg_ws = nullptr; h->cg_ws_floats = 0;
        SCHK(h, hipMalloc((void**)&h->cg_ws, 2 * A * 4));
        h->cg_ws_floats = 2 * A;
    }
    if (!h->ev_host) SCHK(h, hipEventCreateWithFlags(&h->ev_host, hipEventDisableTiming));
    rnde_status st = nsde_head_reserve(h, B, n_classes);   // (the hook runs between an event record and the host's wait on it: it only enqueues)
    if (st != RNDE_OK) return st;
    float* u = h->cg_ws; float* ubar = h->cg_ws + A;
    h->cg_sv.resize((size_t)h->cfg.max_attempts + 1);
    int32_t nsv = 0;
    h->after_solve = [&](hipStream_t s) -> rnde_status {
        return rnde_nsde_classifier_head(h, u, p3_dev, y_dev, B, n_classes, nullptr, ubar, p3_bar_dev, ce_out_dev, s);
    };
    st = nsde_forward_impl(h, x_dev, p2_dev, B, t0, t1, noise_dev, n_pool, seed, nullptr, 0, u, nfe1_out, nfe2_out, h->cg_sv.data(), &nsv, 1, stream);
    h->after_solve = nullptr;
    if (st != RNDE_OK) return st;
    double reg = 0.0;
    const bool regularize = lambda != 0.f && nsv > 0 && h->cfg.regularize != RNDE_REG_NONE;
    if (regularize) {   // lambda * mean(sv.saveval): every saved value carries the cotangent lambda / n
        for (int i = 0; i < nsv; ++i) reg += h->cg_sv[i];
        reg = (double)lambda * reg / nsv;
        for (int i = 0; i < nsv; ++i) h->cg_sv[i] = lambda / (float)nsv;
    }
    if (reg_out_host) *reg_out_host = (float)reg;
    return nsde_backward_impl(h, ubar, regularize ? h->cg_sv.data() : nullptr, x_bar_dev, p2_bar_dev, stream, false);
}

extern "C" rnde_status rnde_nsde_timing(rnde_nsde* h, float* solve_ms, float* rev_sweep_ms, int32_t* attempts, int32_t* accepted) {
    if (!h) return RNDE_ERR_BAD_ARG;
    float a = -1.f, b = -1.f;
    if (h->tev_f) { SCHK(h, hipEventSynchronize(h->tev[1])); SCHK(h, hipEventElapsedTime(&a, h->tev[0], h->tev[1])); }
    if (h->tev_b) { SCHK(h, hipEventSynchronize(h->tev[3])); SCHK(h, hipEventElapsedTime(&b, h->tev[2], h->tev[3])); }
    if (solve_ms) *solve_ms = a;
    if (rev_sweep_ms) *rev_sweep_ms = b;
    if (attempts) *attempts = h->n_att;
    if (accepted) *accepted = h->n_acc;
    return RNDE_OK;
}

extern "C" rnde_status rnde_normal_fill(float* out_dev, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
    if (!out_dev || n < 0) return RNDE_ERR_BAD_ARG;
    if (n == 0) return RNDE_OK;
    hipLaunchKernelGGL(rnde_normal_fill_kernel, dim3((unsigned)std::min<long long>((n / 4 + 255) / 256 + 1, 4096)), dim3(256), 0, (hipStream_t)stream, out_dev,
                       (long long)n, (unsigned long long)seed, (unsigned long long)stream_id);
    return hipGetLastError() == hipSuccess ? RNDE_OK : RNDE_ERR_HIP;
}
