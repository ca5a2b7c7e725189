// rnde_latent.hip -- C ABI of the latent-ODE caller of the hot path (include/rnde.h: rnde_latent_*) over the kernels of rnde_latent.h.
// No torch, no oracle, no CPU fallback.
#include "../../include/rnde.h"
#include "rnde_latent.h"
#include "rnde_device.h"      // DeviceOnce

#include <algorithm>
#include <cmath>
#include <string>

constexpr int kFusedSegs = 16;             // segments of the first reduction level over those partials
constexpr int kFusedWgradGroups = 256;      // workgroups of rnde_latent_gru_wgrad_kernel (one per CU), each leaving one accumulator image

using namespace rnde_lat;

struct rnde_latent {
    rnde_latent_config cfg{};
    float *act = nullptr, *del = nullptr, *y = nullptr, *yb = nullptr, *h1 = nullptr, *out = nullptr, *d1 = nullptr, *d2 = nullptr;
    float *kl = nullptr, *ll = nullptr, *gD = nullptr, *eps = nullptr;
    float *raw = nullptr, *raw2 = nullptr, *raw_side = nullptr, *raw2_side = nullptr;      // accumulator images of the fused weight-gradient passes (main stream: up to 144 tiles; side stream: 28)
    int B = 0, T = 0;
    bool encoded = false;
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // rec_to_gen's weight gradients run beside the reverse GRU (32 workgroups)
    std::string err;
};
static thread_local std::string g_latent_err;

#define LCHK(h, call)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e__); return RNDE_ERR_HIP; } \
    } while (0)

extern "C" const char* rnde_latent_last_error(const rnde_latent* h) { return h ? h->err.c_str() : g_latent_err.c_str(); }
extern "C" void rnde_latent_param_counts(int32_t* n_p1, int32_t* n_p2, int32_t* n_p4) {
    if (n_p1) *n_p1 = kP1;
    if (n_p2) *n_p2 = kP2;
    if (n_p4) *n_p4 = kP4;
}

extern "C" rnde_status rnde_latent_create(const rnde_latent_config* c, rnde_latent** out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c->device) { g_latent_err = "no HIP device"; return RNDE_ERR_NO_DEVICE; }
    if (c->max_batch < 1 || c->max_T < 1 || c->max_T > 64) { g_latent_err = "max_batch >= 1, 1 <= max_T <= 64 (one thread per save time in the likelihood kernel)"; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { g_latent_err = "hipSetDevice failed"; return RNDE_ERR_HIP; }
    rnde_latent* h = new rnde_latent();
    h->cfg = *c;
    const size_t S = (size_t)c->max_batch * c->max_T, B = c->max_batch;
    bool ok = hipMalloc((void**)&h->act, S * kActLd * 4) == hipSuccess && hipMalloc((void**)&h->del, S * kDelLd * 4) == hipSuccess &&
              hipMalloc((void**)&h->y, B * 2 * kL * 4) == hipSuccess && hipMalloc((void**)&h->yb, B * 2 * kL * 4) == hipSuccess &&
              hipMalloc((void**)&h->h1, B * kRec * 4) == hipSuccess && hipMalloc((void**)&h->out, B * 2 * kLat * 4) == hipSuccess &&
              hipMalloc((void**)&h->d1, B * kRec * 4) == hipSuccess && hipMalloc((void**)&h->d2, B * 2 * kLat * 4) == hipSuccess &&
              hipMalloc((void**)&h->kl, B * 4) == hipSuccess && hipMalloc((void**)&h->ll, B * 4) == hipSuccess &&
              hipMalloc((void**)&h->gD, S * 40 * 4) == hipSuccess && hipMalloc((void**)&h->eps, B * kLat * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&h->raw, (size_t)kFusedWgradGroups * kFwWaves * kFwTilesPerWave * 256 * 4) == hipSuccess && hipMalloc((void**)&h->raw2, (size_t)kFusedSegs * kFwWaves * kFwTilesPerWave * 256 * 4) == hipSuccess &&
         hipMalloc((void**)&h->raw_side, (size_t)kFusedWgradGroups * 32 * 256 * 4) == hipSuccess && hipMalloc((void**)&h->raw2_side, (size_t)kFusedSegs * 32 * 256 * 4) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess;
    if (!ok) { g_latent_err = "device allocation failed"; rnde_latent_destroy(h); return RNDE_ERR_HIP; }
    *out = h;
    return RNDE_OK;
}
extern "C" void rnde_latent_destroy(rnde_latent* h) {
    if (!h) return;
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (float* p : {h->act, h->del, h->y, h->yb, h->h1, h->out, h->d1, h->d2, h->kl, h->ll, h->gD, h->eps, h->raw, h->raw2, h->raw_side, h->raw2_side}) if (p) (void)hipFree(p);
    delete h;
}

static rnde_status check_shape(rnde_latent* h, int B, int T) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (B < 1 || B > h->cfg.max_batch || T < 1 || T > h->cfg.max_T) { h->err = "B or T outside the handle's limits"; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return RNDE_ERR_HIP; }
    return RNDE_OK;
}

// out (M x (N + 1), Flux layout [vec(W); b]) = sum over K samples of delta^T [act, 1]: the jobs of one tape pair
struct JobList {
    WgradJobs jj{};
    int fused_groups = 0;      // workgroups of the pass (each leaves one partial)
    void add(const float* delta, int M, const float* act, int N, float* out, int m_split = 1 << 30, int m_gap = 0) {
        jj.j[jj.n++] = WgradJob{delta, act, out, M, N, m_split, m_gap};
    }
};

// ONE pass over two tapes of whole records (rnde_latent_gru_wgrad_kernel): partials per workgroup, then the fixed-order reduction in two levels
// (kFusedSegs segments of the workgroups, then the segments + placement).
template <int LDA, int LDD>
static rnde_status run_fused(rnde_latent* h, const JobList& Jl, const float* act, const float* del, int K, hipStream_t s, float* raw, float* raw2) {
    FusedWgrad F{Jl.jj, act, del, K, raw};
    const int G = Jl.fused_groups, E = fused_tile_count(Jl.jj) * 256;
    constexpr size_t lds = sizeof(float) * 2 * (kFwSamples * (size_t)(LDA + LDD) + 16);      // two images
    static DeviceOnce attr;
    if (attr.need() && lds > 64 * 1024) { LCHK(h, hipFuncSetAttribute((const void*)rnde_latent_gru_wgrad_kernel<LDA, LDD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr.done(); }
    hipLaunchKernelGGL((rnde_latent_gru_wgrad_kernel<LDA, LDD>), dim3(G), dim3(64 * kFwWaves), lds, s, F);
    const int nseg = std::min(kFusedSegs, G), per = (G + nseg - 1) / nseg, segs = (G + per - 1) / per;
    hipLaunchKernelGGL(rnde_latent_reduce_raw_kernel, dim3(std::min((E / 4 + 255) / 256, 64), segs), dim3(256), 0, s, (const float*)raw, G, per, E, raw2);
    hipLaunchKernelGGL(rnde_latent_reduce_scatter_kernel, dim3((E + 255) / 256), dim3(256), 0, s, F, (const float*)raw2, segs, E);
    LCHK(h, hipGetLastError());
    return RNDE_OK;
}

extern "C" rnde_status rnde_latent_encode(rnde_latent* h, const float* x_dev, const float* p1_dev, const float* p2_dev, const float* eps_dev, int32_t B,
                                          int32_t T, float* z0_out_dev, float* mu0_out_dev, float* logvar_out_dev, void* stream) {
    rnde_status st = check_shape(h, B, T);
    if (st != RNDE_OK) return st;
    if (!x_dev || !p1_dev || !p2_dev || !eps_dev || !z0_out_dev || !mu0_out_dev || !logvar_out_dev) { h->err = "null pointer"; return RNDE_ERR_BAD_ARG; }
    hipStream_t s = (hipStream_t)stream;
    h->B = B; h->T = T; h->encoded = false;
    GruParams G{x_dev, p1_dev, h->act, h->del, h->y, B, T};
    const size_t lds = sizeof(float) * (31 * 256 + (size_t)T * 16);
    hipLaunchKernelGGL(rnde_latent_gru_fwd_kernel, dim3((B + 15) / 16), dim3(512), lds, s, G);
    LCHK(h, hipGetLastError());
    LCHK(h, hipMemcpyAsync(h->eps, eps_dev, (size_t)B * kLat * 4, hipMemcpyDeviceToDevice, s));      // the tape owns its copy of the sample
    EncParams E{h->y, p2_dev, h->eps, h->h1, h->out, z0_out_dev, mu0_out_dev, logvar_out_dev, h->kl, B};
    hipLaunchKernelGGL(rnde_latent_enc_fwd_kernel, dim3(B), dim3(64), 0, s, E);
    LCHK(h, hipGetLastError());
    h->encoded = true;
    return RNDE_OK;
}

extern "C" rnde_status rnde_latent_decode_loss(rnde_latent* h, const float* res_dev, const float* p4_dev, const float* x_dev, int32_t B, int32_t T,
                                               float* loss2_out_dev, float* res_bar_out_dev, float* p4_bar_out_dev, void* stream) {
    rnde_status st = check_shape(h, B, T);
    if (st != RNDE_OK) return st;
    if (!h->encoded || B != h->B || T != h->T) { h->err = "decode_loss follows rnde_latent_encode of the same batch (the KL term comes from it)"; return RNDE_ERR_NO_TAPE; }
    if (!res_dev || !p4_dev || !x_dev || !loss2_out_dev || !res_bar_out_dev || !p4_bar_out_dev) { h->err = "null pointer"; return RNDE_ERR_BAD_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const float sigma = 0.01f;      // latent_ode.jl:196
    DecParams D{res_dev, p4_dev, x_dev, h->gD, res_bar_out_dev, h->ll, B, T, 1.0f / (sigma * sigma), -logf(sigma) - 0.5f * logf(2.0f * 3.14159265358979323846f)};
    hipLaunchKernelGGL(rnde_latent_dec_loss_kernel, dim3(B), dim3(64), 0, s, D);
    hipLaunchKernelGGL(rnde_latent_loss_kernel, dim3(1), dim3(256), 0, s, h->ll, h->kl, B, loss2_out_dev);
    LCHK(h, hipGetLastError());
    JobList Jl;
    Jl.fused_groups = std::min(kFusedWgradGroups, (B * T + kFwSamples - 1) / kFwSamples);
    Jl.add(h->gD, kIn, res_dev, kLat, p4_bar_out_dev);      // gen_to_data: [vec(W4) (37 x 20); b4]
    return run_fused<kLat, 40>(h, Jl, res_dev, h->gD, B * T, s, h->raw, h->raw2);
}

extern "C" rnde_status rnde_latent_encode_backward(rnde_latent* h, const float* z0_bar_dev, float lambda_k, const float* p1_dev, const float* p2_dev,
                                                   const float* x_dev, float* p1_bar_out_dev, float* p2_bar_out_dev, void* stream) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (!h->encoded) { h->err = "encode_backward without rnde_latent_encode"; return RNDE_ERR_NO_TAPE; }
    if (!z0_bar_dev || !p1_dev || !p2_dev || !x_dev || !p1_bar_out_dev || !p2_bar_out_dev) { h->err = "null pointer"; return RNDE_ERR_BAD_ARG; }
    if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return RNDE_ERR_HIP; }
    hipStream_t s = (hipStream_t)stream;
    const int B = h->B, T = h->T, K = B * T;
    EncBwdParams E{z0_bar_dev, p2_dev, h->eps, h->h1, h->out, h->d2, h->d1, h->yb, lambda_k / (float)B, B};
    hipLaunchKernelGGL(rnde_latent_enc_bwd_kernel, dim3(B), dim3(128), 0, s, E);
    LCHK(h, hipGetLastError());
    // rec_to_gen: Dense(100, 50, tanh) [W1; b1] then Dense(50, 40) [W2; b2]
    rnde_status st;
    {   // (one job per pass: the two layers' tapes are separate arrays)  These launches do not feed the reverse GRU, which is a 173 us latency chain on
        // 32 workgroups: they run on a stream of their own beside it and are joined in front of the GRU's own weight gradients
        LCHK(h, hipEventRecord(h->ev_fork, s));
        LCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        JobList Je1, Je2;
        Je1.fused_groups = Je2.fused_groups = std::min(kFusedWgradGroups, (B + kFwSamples - 1) / kFwSamples);
        Je1.add(h->d1, kRec, h->y, 2 * kL, p2_bar_out_dev);
        if ((st = run_fused<2 * kL, kRec>(h, Je1, h->y, h->d1, B, h->side, h->raw_side, h->raw2_side)) != RNDE_OK) return st;
        Je2.add(h->d2, 2 * kLat, h->h1, kRec, p2_bar_out_dev + 2 * kL * kRec + kRec);
        if ((st = run_fused<kRec, 2 * kLat>(h, Je2, h->h1, h->d2, B, h->side, h->raw_side, h->raw2_side)) != RNDE_OK) return st;
        LCHK(h, hipEventRecord(h->ev_join, h->side));
    }
    GruParams G{x_dev, p1_dev, h->act, h->del, h->yb, B, T};
    const size_t lds = sizeof(float) * (23 * 256 + (size_t)T * 16);
    hipLaunchKernelGGL(rnde_latent_gru_bwd_kernel, dim3((B + 15) / 16), dim3(512), lds, s, G);
    LCHK(h, hipGetLastError());
    LCHK(h, hipStreamWaitEvent(s, h->ev_join, 0));
    // the six Dense layers of the GRU, in Flux.destructure order: update_gate (Wu1, Wu2), reset_gate (Wr1, Wr2), new_state (Wn1, Wn2): one launch
    float* g = p1_bar_out_dev;
    JobList Jg;
    Jg.fused_groups = std::min(kFusedWgradGroups, (K + kFwSamples - 1) / kFwSamples);
    Jg.add(h->del + dZU, kH, h->act + aYC, kNIn, g + oWu1);
    Jg.add(h->del + dAU, kL, h->act + aU1, kH, g + oWu2);
    Jg.add(h->del + dZR, kH, h->act + aYC, kNIn, g + oWr1);
    Jg.add(h->del + dAR, kL, h->act + aR1, kH, g + oWr2);
    Jg.add(h->del + dZN, kH, h->act + aCC, kNIn, g + oWn1);
    Jg.add(h->del + dNS, 2 * kL, h->act + aN1, kH, g + oWn2, kL, 2);
    if ((st = run_fused<kActLd, kDelLd>(h, Jg, h->act, h->del, K, s, h->raw, h->raw2)) != RNDE_OK) return st;      // all six in one pass over the tapes
    h->encoded = false;
    return RNDE_OK;
}

// Optimiser(InvDecay(gamma), AdaMax(eta, (beta1, beta2))) step of one flat parameter group (reference experiments/latent_ode.jl:108).
// n: the group's InvDecay counter (1 at the first step), beta1_pow: beta1^t of Flux's running state (beta1 at the first step); the caller
// advances both.  Asynchronous on `stream`.
extern "C" rnde_status rnde_adamax_step(float* p_dev, const float* g_dev, float* m_dev, float* u_dev, int64_t len, int64_t n, float gamma, float eta,
                                        float beta1, float beta2, float eps, float beta1_pow, void* stream) {
    if (!p_dev || !g_dev || !m_dev || !u_dev || len < 0 || !(beta1_pow < 1.f)) return RNDE_ERR_BAD_ARG;
    if (len == 0) return RNDE_OK;
    const unsigned blocks = (unsigned)((len + 255) / 256 > 1024 ? 1024 : (len + 255) / 256);
    hipLaunchKernelGGL(rnde_adamax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p_dev, g_dev, m_dev, u_dev, (long long)len,
                       1.0f / (1.0f + gamma * (float)n), eta / (1.0f - beta1_pow), beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? RNDE_OK : RNDE_ERR_HIP;
}
