// rnde_reverse.hip -- C ABI (include/rnde.h), the reverse side: discretise-then-optimise reverse passes of the three ODE engines, the weight-gradient
// GEMMs, the classifier head and its fused training step, the optimiser steps.  The handle and what rnde.hip offers this file: rnde_node.h.
#include "rnde_node.h"
#include "rnde_wgradx.h"

static rnde_status chain_bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                                 float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync = true, float* tspan_bar_dev = nullptr);
static rnde_status bwd_run(rnde_node* h, const float* u_bar_dev, const float* saveval_bar_host, float* x_bar_dev,
                           float* p_bar_dev, float* tspan_bar_host, hipStream_t s, bool sync = true, float* tspan_bar_dev = nullptr);


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

// ---- reverse pass driver ------------------------------------------------------------------------
static rnde_status bwd_prepare(rnde_node* h) {
    BwdBuffers& b = h->bw;
    if (b.ready) return RNDE_OK;
    const size_t A = (size_t)h->D * h->Bpad_max, HB = (size_t)h->H * h->Bpad_max;
    const int cap = h->cfg.max_attempts;
    HIPCHK(h, hipMalloc((void**)&b.U, A * 4)); HIPCHK(h, hipMalloc((void**)&b.K1, A * 4)); HIPCHK(h, hipMalloc((void**)&b.UB1, A * 4));
    HIPCHK(h, hipMalloc((void**)&b.zi2, 2 * A * 4)); HIPCHK(h, hipMalloc((void**)&b.zi1, 2 * HB * 4));
    HIPCHK(h, hipMalloc((void**)&b.svb_att, (size_t)cap * 4));
    HIPCHK(h, hipMalloc((void**)&b.bstate, 2 * sizeof(BState) + 64)); HIPCHK(h, hipMalloc((void**)&b.ibstate, 2 * sizeof(IBState)));      // (+ 64 bytes: the [2][4] sums of rnde_bpart_reduce_kernel)
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
        // matrix mode 1 in effect for this step (the forward ran the x3 solve): the GEMMs on the matrix cores too (rnde_wgradx.h; RNDE_X3_WGRAD_OFF=1: A/B)
        const bool x3_off = getenv("RNDE_X3_WGRAD_OFF") != nullptr;            // (read per call: A/B runs and tests switch inside one process)
        const bool x3_half = getenv("RNDE_X3_WGRAD_HALF") != nullptr;         // (A/B: the single-buffered half form)
        const int TT4 = ((tall ? M : Nx + 2) + 15) / 16;
        if (h->x3_packed && !x3_off && !x3_half && TT4 >= 28 && TT4 <= 52) {
            // quarter form (rnde_wgrad4x_kernel): four workgroups per chunk, two LDS images; the launches underneath the sweep stay at 32 workgroups (8 chunks)
            static const int target4 = getenv("RNDE_WGRAD4_CHUNKS") ? atoi(getenv("RNDE_WGRAD4_CHUNKS")) : 64;
            int sc4 = std::max(1, std::min({max_chunks > 0 ? max_chunks / 2 : target4, total_steps, 256}));
            const int spc4 = (total_steps + sc4 - 1) / sc4;
            sc4 = (total_steps + spc4 - 1) / spc4;
            if ((size_t)(*chunk_cursor + sc4) * (size_t)len > h->bw.slab_floats) { h->err = "weight-gradient slab overflow"; return RNDE_ERR_BAD_ARG; }
            static DeviceOnce attr4;
            if (attr4.need()) {
                HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad4x_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWx4LdsBytes));
                HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad4x_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWx4LdsBytes));
                attr4.done();
            }
            const dim3 g4(32 * ((sc4 + 7) / 8));
            if (tall) hipLaunchKernelGGL((rnde_wgrad4x_kernel<true>), g4, dim3(448), kWx4LdsBytes, s, ev, n_evals, spc4, sc4, M, Nx, Bpad, dst);
            else hipLaunchKernelGGL((rnde_wgrad4x_kernel<false>), g4, dim3(448), kWx4LdsBytes, s, ev, n_evals, spc4, sc4, M, Nx, Bpad, dst);
            HIPCHK(h, hipGetLastError());
            *chunk_cursor += sc4;
            return RNDE_OK;
        }
        if (h->x3_packed && !x3_off) {
            static DeviceOnce attrx;
            if (attrx.need()) {
                HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad3x_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWxLdsBytes));
                HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad3x_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWxLdsBytes));
                attrx.done();
            }
            if (tall) hipLaunchKernelGGL((rnde_wgrad3x_kernel<true>), dim3(2, sc), dim3(448), kWxLdsBytes, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
            else hipLaunchKernelGGL((rnde_wgrad3x_kernel<false>), dim3(2, sc), dim3(448), kWxLdsBytes, s, ev, n_evals, steps_per_chunk, M, Nx, Bpad, dst);
            HIPCHK(h, hipGetLastError());
            *chunk_cursor += sc;
            return RNDE_OK;
        }
        const size_t lds = (size_t)2 * 32 * (464 + 144) * sizeof(float);   // two buffers
        static DeviceOnce attr;
        if (attr.need()) {
            HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(h, hipFuncSetAttribute((const void*)rnde_wgrad3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr.done();
        }
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
        BQ.B = Q; BQ.p = h->pcopy; BQ.pwBt = h->spwBt; BQ.pwDt = h->spwDt; BQ.slab = h->slab2; BQ.x3Bt = h->x3Bt; BQ.x3Dt = h->x3Dt;
        BQ.UTB = b.UTB; BQ.UNB = b.UNB; BQ.UPB0 = b.UPB0; BQ.GB = b.GB;
        BQ.EXK = b.GB + 6 * A; BQ.EXG = b.GB + 7 * A; BQ.SVW = b.GB + 8 * A;
        BQ.sv_t = h->saveat.empty() ? nullptr : h->sv_t_dev; BQ.sv_ubar = u_bar_dev; BQ.nsave = (int)h->saveat.size();
        BQ.MT = h->sMT; BQ.WT = h->sWT; BQ.R = h->sR; BQ.C = Q.F.Bpad / 16; BQ.HT = h->sHT; BQ.KHb = h->sKHb;
        // large batches: the partials of attempt n + 1 are summed ONCE behind its launch (rnde_bpart_reduce_kernel) instead of in the START of every
        // workgroup of attempt n -- pays from ~900 partials on (B >= 2048: a 4 us launch against 7+ us of dependent cold loads in every workgroup)
        double* const bsum = (double*)(b.bstate + 2);
        const bool pre_reduce = h->persist == 1 && Q.bpart_n >= 896 && !getenv("RNDE_NO_BPART_REDUCE");
        if (pre_reduce) BQ.B.bsum = bsum;
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
            if (h->cfg.regularize == RNDE_REG_STIFF_DT && eg_ok) eigb = (double)b.h_svb[n] * ((double)mm.eigen * (double)mm.dt > 0 ? 1.0 : -1.0) * (double)mm.dt;      // |eigen_est * dt| (test/test_node.jl:75)
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
                    static DeviceOnce attr;
                    if (attr.need()) {
                        HIPCHK(h, hipFuncSetAttribute((const void*)rnde_bstage_attempt_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        HIPCHK(h, hipFuncSetAttribute((const void*)rnde_bstage_attempt_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        attr.done();
                    }
                    // matrix mode 1 (rnde_x3.h): the forward ran the x3 solve and split the transposed weights too; the x3 form serves every callback of the
                    // experiments (the eigen_est cotangents fit since the partial sums became scalars: 198 VGPRs), and saveat in an instantiation of its own (256)
                    const bool x3 = h->x3_packed && h->x3Bt && h->x3Dt && !getenv("RNDE_X3_REV_OFF");
                    if (x3) {
                        const size_t xlds = sizeof(float) * ((size_t)2 * kX3ImageFloats + 64) + (size_t)(RNDE_BSTAGE_HDMA ? 1 : 0) * 6 * 7 * 2 * 1024;
                        static DeviceOnce attr3;
                        if (attr3.need()) {
#define RNDE_X3_ATTR(A, E) HIPCHK(h, hipFuncSetAttribute((const void*)rnde_bstage_attempt_kernel<A, 1, 1, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                            RNDE_X3_ATTR(1, 0) RNDE_X3_ATTR(1, 1) RNDE_X3_ATTR(1, 2) RNDE_X3_ATTR(1, 3) RNDE_X3_ATTR(0, 0) RNDE_X3_ATTR(0, 1) RNDE_X3_ATTR(0, 2) RNDE_X3_ATTR(0, 3)
#undef RNDE_X3_ATTR
                            attr3.done();
                        }
                        // the instantiation by what this reverse pass needs: saveat cotangents (bit 0), eigen_est cotangents (bit 1: the three stiffness callbacks)
                        const int ex = (h->saveat.empty() ? 0 : 1) | (h->cfg.regularize >= RNDE_REG_STIFF ? 2 : 0);
#define RNDE_X3_LAUNCH(A, E) hipLaunchKernelGGL((rnde_bstage_attempt_kernel<A, 1, 1, E>), pgrid, blk, xlds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n])
                        if (h->act2) { if (ex == 0) RNDE_X3_LAUNCH(1, 0); else if (ex == 1) RNDE_X3_LAUNCH(1, 1); else if (ex == 2) RNDE_X3_LAUNCH(1, 2); else RNDE_X3_LAUNCH(1, 3); }
                        else { if (ex == 0) RNDE_X3_LAUNCH(0, 0); else if (ex == 1) RNDE_X3_LAUNCH(0, 1); else if (ex == 2) RNDE_X3_LAUNCH(0, 2); else RNDE_X3_LAUNCH(0, 3); }
#undef RNDE_X3_LAUNCH
                    } else if (h->act2) hipLaunchKernelGGL((rnde_bstage_attempt_kernel<1, 1>), pgrid, blk, flds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                    else hipLaunchKernelGGL((rnde_bstage_attempt_kernel<0, 1>), pgrid, blk, flds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                } else if (h->act2) hipLaunchKernelGGL((rnde_bstage_attempt_kernel<1, 0>), pgrid, blk, h->stage_lds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                else hipLaunchKernelGGL((rnde_bstage_attempt_kernel<0, 0>), pgrid, blk, h->stage_lds, s, BQ, n, h->h_meta[n], c1, c2, sv_lo[n], sv_hi[n], Y, qo, b.h_svb[n]);
                if ((st = couple_sum(h, b.bpart + (size_t)(n & 1) * Q.bpart_n * 4, 4LL * Q.bpart_n, s)) != RNDE_OK) return st;
                if (pre_reduce && n > 0) hipLaunchKernelGGL(rnde_bpart_reduce_kernel, dim3(1), dim3(64), 0, s, Q, n, bsum);      // (attempt 0's partials go to the reverse of the initial-step rule, which sums them itself)
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
        fprintf(stderr, "  total %lld cycles to the end of stage 1, END (reduction of the 21 partial sums) %lld; in START's vector part the wait for the record's k arrays %lld\n",
                (long long)(hst[34]-hst[0]), (long long)(hst[35]-hst[34]), (long long)(hst[45]-hst[42]));
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
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_bchain_kernel<NKD, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchain_init_kernel<NKD, 1, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchain_init_kernel<NKD, 2, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.done();
    }
    const dim3 grid(Q.B.F.nwg), blk(64 * kCW);
    for (int n = Q.B.n_att - 1; n >= 0; --n) {
        float c1 = 0.f, c2 = 0.f;   // cotangent of eigen_est for this attempt (as in bwd_run)
        const StepMeta& mm = h->h_meta[n];
        const bool eg_ok = !(mm.eigen == 0.f || mm.eigen != mm.eigen);
        double eigb = 0.0;
        if (h->cfg.regularize == RNDE_REG_STIFF && eg_ok) eigb = (double)b.h_svb[n] * (mm.eigen > 0 ? 1.0 : -1.0) / 3.5068;
        if (h->cfg.regularize == RNDE_REG_ERR_STIFF && eg_ok) eigb = 0.1 * (double)b.h_svb[n] / 3.5068;
        if (h->cfg.regularize == RNDE_REG_STIFF_DT && eg_ok) eigb = (double)b.h_svb[n] * ((double)mm.eigen * (double)mm.dt > 0 ? 1.0 : -1.0) * (double)mm.dt;      // |eigen_est * dt| (test/test_node.jl:75)
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
    static DeviceOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)rnde_bchainmw_kernel<NR, TAB, LAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_kernel<NR, TAB, LAT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_init_kernel<NR, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rnde_bchainmw_init_kernel<NR, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        HIPCHK(h, e);
        attr_set.done();
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
        if (h->cfg.regularize == RNDE_REG_STIFF_DT && eg_ok) eigb = (double)b.h_svb[n] * ((double)mm.eigen * (double)mm.dt > 0 ? 1.0 : -1.0) * (double)mm.dt;      // |eigen_est * dt| (test/test_node.jl:75)
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
        h->pending_bsweep = true; h->bsweep_nt = Q.ntiles; h->bsweep_global = W.xch_global != 0;
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

