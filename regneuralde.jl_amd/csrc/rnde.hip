// rnde.hip -- C ABI (include/rnde.h) over the gfx950 kernels.  No torch, no oracle, no CPU fallback:
// every entry point either runs the HIP kernels or returns an error status.
#include "rnde_node.h"

static thread_local std::string g_create_err;   // last create error of the calling thread (rnde_last_error(NULL)); no process-wide mutable state
extern "C" const char* rnde_version(void) { return "rnde 0.1.0 (gfx950)"; }
extern "C" const char* rnde_last_error(const rnde_node* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int32_t rnde_param_count(const rnde_node_config* c) {
    int n = 0;
    for (int l = 0; l < c->n_layers; ++l) n += (c->dims[l] + (c->time_dep ? 1 : 0)) * c->dims[l + 1] + c->dims[l + 1];
    return n;
}

StepParams make_params(rnde_node* h, const float* x, int B, float t0, float t1, int tape) {
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
    if (c->regularize < RNDE_REG_NONE || c->regularize > RNDE_REG_STIFF_DT) { g_create_err = "regularize: unknown value"; return RNDE_ERR_BAD_ARG; }
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
    ok &= dm((void**)&h->errpart, (size_t)(6 * h->nwg_max + 256 + 16) * 4) && dm((void**)&h->initpart, (size_t)(3 * h->nwg_max + 256) * 4);   // (+256: sum_partials reads whole 256-entry blocks; +16: the [2][4] doubles of rnde_epart_reduce_kernel)
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
ChainParams make_chain_params(rnde_node* h, const StepParams& P) {
    ChainParams Q{};
    Q.F = P; Q.G = h->cg; Q.frags = h->cfrags; Q.ntiles = P.Bpad / 16;
    return Q;
}
template <int NKD, int MODE, int ALT = 0>
static hipError_t launch_chain_t(rnde_node* h, const ChainParams& Q, int n, float* u_out, hipStream_t s) {
    auto kern = rnde_chain_kernel<NKD, MODE, ALT>;
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.done();
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
MwParams make_mw_params(rnde_node* h, const StepParams& P) {
    MwParams Q{};
    Q.F = P; Q.G = h->mg; Q.rk = h->rk; Q.tab = h->mw_tab; Q.ntiles = P.Bpad / 16; Q.u_out = nullptr;
    Q.ev_stride = (long long)Q.ntiles * h->mg.RS * 64;
    Q.slab = P.tape ? h->mw_slab : nullptr;
    return Q;
}
template <int NR, int MODE, int TAB, int LAT = 0>
static hipError_t launch_mw_t(rnde_node* h, const MwParams& Q, int n, hipStream_t s) {
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_chainmw_kernel<NR, MODE, TAB, LAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.done();
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
    long long want = std::max(evals, 2 * h->mw_slab_evals);
    {   // 288 GB of HBM: when the slab of a max_attempts solve is small next to what is free, take that at once -- growing it later costs a stream
        // synchronisation, a multi-GB hipMalloc, a copy and a hipFree (~40 ms), and while a model trains its step count creeps across the doubling
        // thresholds in the middle of a run (seen in bench.py's latent_e2e record: one 45 ms step among 3.5 ms ones).  Bounded: the memory is invisible to
        // the caller's own allocator (torch's caching allocator), several handles per layer and several ranks may share one device -- at most
        // RNDE_EAGER_SLAB_MB (default 8192: the config-4 slab at B = 512, max_attempts 256 is 6.8 GB) and at most an eighth of what is free; beyond that the slab grows by doubling as before.
        const long long full = 2 + (long long)(h->rk_S - 1) * h->cfg.max_attempts;
        long long cap_mb = 8192;
        if (const char* e = getenv("RNDE_EAGER_SLAB_MB")) cap_mb = atoll(e);
        size_t fr = 0, tot = 0;
        const size_t full_bytes = (size_t)full * per_eval * 4;
        if (full > want && cap_mb > 0 && full_bytes <= (size_t)cap_mb << 20 && hipMemGetInfo(&fr, &tot) == hipSuccess && full_bytes <= fr / 8) want = full;
    }
    float* nb = nullptr;
    HIPCHK(h, hipStreamSynchronize(s));
    if (hipMalloc((void**)&nb, (size_t)want * per_eval * 4) != hipSuccess) {      // the eager size did not fit after all (fragmentation, a neighbour): what the call needs, no more
        (void)hipGetLastError();
        nb = nullptr;
        want = evals;
        HIPCHK(h, hipMalloc((void**)&nb, (size_t)want * per_eval * 4));
    }
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
    if (c->regularize < RNDE_REG_NONE || c->regularize > RNDE_REG_STIFF_DT) { g_create_err = "regularize: unknown value"; return RNDE_ERR_BAD_ARG; }
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
    ok &= dm((void**)&h->errpart, (size_t)(6 * h->nwg_max + 256 + 16) * 4) && dm((void**)&h->initpart, (size_t)(3 * h->nwg_max + 256) * 4);   // (+256: sum_partials reads whole 256-entry blocks; +16: the [2][4] doubles of rnde_epart_reduce_kernel)
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
    // the stage kernels' Dense layers on the matrix cores (rnde_x3.h): split weight images for the headline geometry (49 row tiles, 7 x 7 (hidden tile, row
    // block) pairs) whenever the one-launch-per-attempt kernels are in use (persist == 1) -- with or without the one-launch solve
    if (h->persist == 1) {
        h->x3 = 1;      // default where the geometry fits (include/rnde.h: rnde_node_set_matrix_mode)
        if (const char* e7 = getenv("RNDE_X3")) h->x3 = atoi(e7) != 0;
        if (h->sMT == 49 && h->sHT == 7 && h->sR == 7 && h->sWT == 7 && h->D == 784 && h->H == 100 && !h->stage_generic) {
            const size_t img = (size_t)49 * 4 * 3 * 64 * 16;
            if (hipMalloc(&h->x3B, img) != hipSuccess || hipMalloc(&h->x3D, img) != hipSuccess || hipMalloc(&h->x3Bt, img) != hipSuccess || hipMalloc(&h->x3Dt, img) != hipSuccess) { g_create_err = "device allocation failed"; rnde_node_destroy(h); return RNDE_ERR_HIP; }
        } else h->x3 = 0;
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
    if (h->x3B) hipFree(h->x3B);
    if (h->x3D) hipFree(h->x3D);
    if (h->x3Bt) hipFree(h->x3Bt);
    if (h->x3Dt) hipFree(h->x3Dt);
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
rnde_status couple_sum(rnde_node* h, float* partials, long long count, hipStream_t s) {
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
hipError_t stage_pack(rnde_node* h, const float* p, f32x4* dst, int which, int MTrows, int Kb, hipStream_t st) {
    const long long total = (long long)MTrows * Kb * 64;
    const int grid = (int)std::min<long long>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(rnde_stage_pack_kernel, dim3(grid), dim3(256), 0, st, p, dst, which, h->D, h->H, MTrows, Kb);
    return hipGetLastError();
}
// matrix mode 1 (rnde_x3.h): the weights split into three bf16 planes for the x3 kernels, in front of every forward (p changes between training steps);
// with_rev: the transposed pair of the reverse attempt kernel in the same launch.  Sets "this forward's stage kernels run with x3".
static rnde_status x3_pack(rnde_node* h, const float* p_dev, bool with_rev, hipStream_t s) {
    h->x3_fwd = false;
    if (!(h->x3 && h->x3B && h->x3D && h->persist == 1 && !h->stage_generic)) return RNDE_OK;
    HIPCHK(h, rnde_launch_x3_pack(p_dev, h->x3B, h->x3D, with_rev ? h->x3Bt : nullptr, with_rev ? h->x3Dt : nullptr, h->D, h->H, h->sMT, h->sWT, h->sR, h->sHT, s));
    h->x3_fwd = true;
    h->x3_packed = with_rev;
    return RNDE_OK;
}
static rnde_status stage_pack_weights(rnde_node* h, const float* p_dev, hipStream_t s) {
    HIPCHK(h, stage_pack(h, p_dev, h->spwB, 0, h->sMT, h->sK2b, s));
    HIPCHK(h, stage_pack(h, p_dev, h->spwD, 1, h->sHT, h->sMT, s));
    return x3_pack(h, p_dev, false, s);
}
rnde_status stage_pack_all(rnde_node* h, const float* p_dev, hipStream_t s, const float* x_src, long long x_floats) {
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
    return x3_pack(h, p_dev, true, s);
}
// The hand-off slabs must read "empty" wherever the persistent kernels have not written in the current tile indexing: at
// creation, and whenever the padded batch width (= number of column tiles) differs from the last persistent launch's.
hipError_t slab_prepare(rnde_node* h, int Bpad, hipStream_t s) {
    if (h->tslab_Bpad == Bpad) return hipSuccess;
    h->tslab_Bpad = Bpad;
    return hipMemsetAsync(h->tslab, 0xFF, h->tslab_bytes, s);
}
// the two-tile attempt kernel serves this geometry (rnde_stage_persist2.h): the headline network, an even number of column tiles, enough of them
static bool stage_two_tile(const rnde_node* h, const StageParams& Q) {
    const bool fix = Q.WT == 7 && Q.HT == 7 && Q.K2b == 7 && Q.MT == 49 && Q.R == 7 && h->D == 784 && h->H == 100 && !h->stage_generic;
    return h->persist == 1 && fix && Q.C % 2 == 0 && h->persist2 != 0 && (Q.C >= kPersist2MinTiles || h->persist2 >= 1);
}
// large batches: the error partials of an attempt summed once behind its launch (rnde_epart_reduce_kernel) instead of in every workgroup's prologue of the next --
// from ~900 partials on (B >= 2048), where the two-tile kernel runs; RNDE_NO_EPART_REDUCE=1: A/B
static bool stage_epart_reduce(const rnde_node* h, const StageParams& Q) {
    return stage_two_tile(h, Q) && Q.F.nwg >= 896 && !getenv("RNDE_NO_EPART_REDUCE");
}
static hipError_t stage_attempt(rnde_node* h, const StageParams& Q, int n, hipStream_t s) {
    if (h->persist == 1) {   // one launch per attempt, slab hand-offs inside the kernel (rnde_stage_persist.h)
        PersistSync Y{h->tslab, h->pabort, h->pxcc, h->persist_spins};
        if (hipError_t e = slab_prepare(h, Q.Bpad16, s); e != hipSuccess) return e;
        const dim3 grid(8 * Q.R * ((Q.C + 7) / 8));   // a column tile's row blocks share blockIdx % 8 (same XCD)
        const bool fix = Q.WT == 7 && Q.HT == 7 && Q.K2b == 7 && Q.MT == 49 && Q.R == 7 && h->D == 784 && h->H == 100 && !h->stage_generic;
        // batches that fill the chip more than once: two column tiles per workgroup (rnde_stage_persist2.h; bit-identical results).
        // RNDE_PERSIST2=0 keeps one tile per workgroup (A/B and the bit-identity test), =1 takes two whenever the tile count is even.
        static const bool x3_over_mt = !(getenv("RNDE_X3_MT") && atoi(getenv("RNDE_X3_MT")) == 0);      // (A/B: 0 = keep the fp32 two-tile kernel for large batches even in matrix mode 1)
        if (stage_two_tile(h, Q)) {
            const dim3 grid2(8 * Q.R * ((Q.C / 2 + 7) / 8));
            if (h->x3_fwd && x3_over_mt) {      // matrix mode 1: the two-tile kernel in its x3 form (RNDE_X3_MT=0: the fp32 form, A/B)
                const size_t xlds2 = sizeof(float) * ((size_t)2 * 2 * kX3ImageFloats + 32 * 3);
                if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<1, 2, 1>), grid2, dim3(64 * 7), xlds2, s, Q, n, Y, (const void*)h->x3B, (const void*)h->x3D);
                else hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<0, 2, 1>), grid2, dim3(64 * 7), xlds2, s, Q, n, Y, (const void*)h->x3B, (const void*)h->x3D);
                return hipGetLastError();
            }
            const size_t lds2 = sizeof(float) * (2 * 2 * 16 * (16 * 7 + 4) + 32 * 3);
            if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<1, 2>), grid2, dim3(64 * 7), lds2, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
            else hipLaunchKernelGGL((rnde_stage_attempt_mt_kernel<0, 2>), grid2, dim3(64 * 7), lds2, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
            return hipGetLastError();
        }
        if (fix && h->x3_fwd) {      // matrix mode 1: the same instruction sequence as the x3 one-launch solve (bit-identical to it)
            const size_t xlds = sizeof(float) * ((size_t)2 * kX3ImageFloats + 64);
            if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_kernel<1, 1, 1>), grid, dim3(64 * 7), xlds, s, Q, n, Y, (const void*)h->x3B, (const void*)h->x3D);
            else hipLaunchKernelGGL((rnde_stage_attempt_kernel<0, 1, 1>), grid, dim3(64 * 7), xlds, s, Q, n, Y, (const void*)h->x3B, (const void*)h->x3D);
        } else if (fix) {
            if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_kernel<1, 1>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
            else hipLaunchKernelGGL((rnde_stage_attempt_kernel<0, 1>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
        } else if (h->act2) hipLaunchKernelGGL((rnde_stage_attempt_kernel<1, 0>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
        else hipLaunchKernelGGL((rnde_stage_attempt_kernel<0, 0>), grid, dim3(64 * Q.WT), h->stage_lds, s, Q, n, Y, (const void*)nullptr, (const void*)nullptr);
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
bool bsweep_failed(rnde_node* h, hipStream_t s) {
    if (!h->pending_bsweep) return false;
    h->pending_bsweep = false;
    bool bad = h->h_mw_bchk[0] != 0;
    const int nt = h->bsweep_nt;      // the sweep's own geometry, recorded at its launch (h->B may already belong to the next forward)
    for (int i = 1; i < nt && !h->bsweep_global && !bad; ++i) bad = h->h_mw_bchk[2 + i] != h->h_mw_bchk[2];      // (memory-side meeting: it does not depend on the placement)
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

// After a synchronisation point: did a persistent launch time out, or did a column tile's workgroups land on different
// XCDs (then their slab hand-off through L2 would not be coherent)?  Either way the persistent kernels are disabled for
// this handle and the caller redoes the solve with the multi-launch kernels.
// (two halves so that the two small copies ride on a synchronisation the caller performs anyway)
void persist_check_enqueue(rnde_node* h, int grid, hipStream_t s) {
    if (h->persist != 1) return;
    h->h_pchk[0] = 1;   // stays 1 ("failed") if a copy cannot even be enqueued
    if (hipMemcpyAsync(h->h_pchk, h->pabort, 8, hipMemcpyDeviceToHost, s) != hipSuccess) return;
    (void)hipMemcpyAsync(h->h_pchk + 2, h->pxcc, (size_t)grid * 4, hipMemcpyDeviceToHost, s);
}
bool persist_check_result(rnde_node* h, int C, int R, hipStream_t s) {   // call after the stream has been synchronised
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

rnde_status forward_impl(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1,
                                float* u_out_dev, const float* saveat_host, int32_t n_saveat, float* sv_out_dev,
                                int64_t* nfe_out, float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    // a solve may ask to be redone for two independent reasons (the record slab has to grow; a one-launch kernel gave up and the handle fell back):
    // both can happen back to back, each at most once per cause -- anything beyond that is an error, not a status the caller should ever see
    rnde_status st = RNDE_INTERNAL_RETRY;
    for (int tries = 0; tries < 4 && st == RNDE_INTERNAL_RETRY; ++tries)
        st = forward_core(h, x_dev, p_dev, B, t0, t1, u_out_dev, saveat_host, n_saveat, sv_out_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    if (st == RNDE_INTERNAL_RETRY) { h->err = "the forward solve asked to be redone four times in a row (one-launch fallbacks and slab growth): giving up"; st = RNDE_ERR_HIP; }
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
    h->have_tape = false; h->rev_packed = false; h->x3_packed = false; h->x3_fwd = false;
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
        if (stage_epart_reduce(h, SQ)) SQ.F.esum = (const double*)(h->errpart + 6 * (size_t)h->nwg_max + 256);      // (inside errpart's allocation, 8-byte aligned)
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
        if (bad) {      // a meeting timed out, or the workgroups did not share an XCD: this handle goes back to one launch per attempt for the next mw_retry_after solves
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
        const int x3 = h->x3_fwd ? 1 : 0;      // (the weights were split by this forward's pack launch: x3_pack)
        SolveSync Z{h->sxch, h->s_epoch, cap, h->x3B, h->x3D};
#ifdef RNDE_DIAG
        StageParams SD = SQ;
        if (getenv("RNDE_DIAG_SOLVE")) {      // cycle stamps of workgroup 0, every attempt (tools/diag_solve.py)
            if (h->diag_buf) hipFree(h->diag_buf);
            hipMalloc((void**)&h->diag_buf, 16384);
            hipMemsetAsync(h->diag_buf, 0, 16384, s);
            SD.F.dbg_out = h->diag_buf;
        }
        HIPCHK(h, rnde_launch_stage_solve(&SD, &Y, &Z, h->act2, x3, s));
#else
        HIPCHK(h, rnde_launch_stage_solve(&SQ, &Y, &Z, h->act2, x3, s));
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
            if (h->engine == 2 && SQ.F.esum) { hipLaunchKernelGGL(rnde_epart_reduce_kernel, dim3(1), dim3(64), 0, s, SQ.F, launched, (double*)SQ.F.esum); HIPCHK(h, hipGetLastError()); }
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
            case RNDE_REG_STIFF_DT: return fabsf(eig * dt);      // test/test_node.jl:75: abs(integrator.eigen_est * integrator.dt)
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

// save_everystep = true (reference src/models/neural_ode.jl:10-11: return_multiple = save_everystep || saveat): the state after every ACCEPTED step.
// Which times those are is known only after the solve, so the call runs it twice -- once untaped for the step sequence, once with the accepted
// step ends (and t0, if asked) as save times: the value saved at a step's end is u_new itself, no interpolation (dense_points, rnde_stage.h),
// and the second run repeats the first bit for bit (save times do not enter the controller).  A call shape for small problems: no reference
// experiment uses it (every multi-output call site passes saveat); the backward call afterwards is the saveat one.
extern "C" rnde_status rnde_node_forward_everystep(rnde_node* h, const float* x_dev, const float* p_dev, int32_t B, float t0, float t1, int32_t save_start,
                                                   float* sol_out_dev, int32_t capacity, float* t_host_out, int32_t* n_out, int64_t* nfe_out,
                                                   float* saveval_host, int32_t* n_saveval_out, int32_t keep_tape, void* stream) {
    if (!h || !n_out || !sol_out_dev || capacity < 1) return RNDE_ERR_BAD_ARG;
    rnde_status st = forward_impl(h, x_dev, p_dev, B, t0, t1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, stream);
    if (st != RNDE_OK) return st;
    std::vector<float> times;
    if (save_start) times.push_back(t0);
    for (int i = 0; i < h->n_att; ++i) {
        const StepMeta& m = h->h_meta[i];
        if (m.flags & F_ACCEPT) { float tn = m.t + m.dt; if (tn > t1) tn = t1; times.push_back(tn); }      // (fp32, as the controller forms t + dt; the last step ends at t1 exactly whenever t >= t1 / 2)
    }
    *n_out = (int32_t)times.size();
    if ((int)times.size() > capacity) { h->err = "rnde_node_forward_everystep: more accepted steps than the output has room for"; return RNDE_ERR_BAD_ARG; }
    if (t_host_out) memcpy(t_host_out, times.data(), times.size() * sizeof(float));
    st = forward_impl(h, x_dev, p_dev, B, t0, t1, nullptr, times.data(), (int32_t)times.size(), sol_out_dev, nfe_out, saveval_host, n_saveval_out, keep_tape, stream);
    if (st != RNDE_OK) return st;
    // "the value at a step's end is u_new itself, no interpolation" holds only if the second solve (another kernel path: it saves) took exactly the steps
    // the first one took -- the same arithmetic, so it must; checked, not assumed (round-4 review)
    size_t k = save_start ? 1 : 0;
    for (int i = 0; i < h->n_att; ++i) {
        const StepMeta& m = h->h_meta[i];
        if (!(m.flags & F_ACCEPT)) continue;
        float tn = m.t + m.dt; if (tn > t1) tn = t1;
        if (k >= times.size() || times[k] != tn) { h->err = "rnde_node_forward_everystep: the saving solve did not retrace the steps of the first one"; return RNDE_ERR_HIP; }
        ++k;
    }
    if (k != times.size()) { h->err = "rnde_node_forward_everystep: the saving solve took fewer accepted steps than the first one"; return RNDE_ERR_HIP; }
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
extern "C" rnde_status rnde_node_set_matrix_mode(rnde_node* h, int32_t mode) {
    if (!h) return RNDE_ERR_BAD_ARG;
    if (mode != RNDE_MATRIX_F32 && mode != RNDE_MATRIX_BF16X3) { h->err = "matrix mode: 0 (fp32-input MFMA) or 1 (bf16x3 on the matrix cores)"; return RNDE_ERR_BAD_ARG; }
    h->x3 = (mode == RNDE_MATRIX_BF16X3 && h->x3B && h->x3D) ? 1 : 0;
    return RNDE_OK;
}
extern "C" int32_t rnde_node_matrix_mode(const rnde_node* h) { return (h && h->x3 && h->x3B && h->x3D) ? RNDE_MATRIX_BF16X3 : RNDE_MATRIX_F32; }
extern "C" int32_t rnde_node_launches_per_attempt(const rnde_node* h) {
    if (!h) return 0;
    return (h->engine == 2 && h->persist != 1) ? 7 : 1;
}

extern "C" rnde_status rnde_node_release_tape(rnde_node* h) {
    if (!h) return RNDE_ERR_BAD_ARG;
    h->have_tape = false;
    return RNDE_OK;
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
