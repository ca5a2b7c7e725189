// rnde_wgradx.h -- the parameter-gradient GEMMs of the stage engine on the MATRIX CORES (the X3 form of rnde_wgrad3_kernel, rnde_bwd.h).
//
//   layer 2:  W2x-bar (784 x 102) = sum over evaluations e, batch columns b of  z2-bar_e[:, b] [h_e; t_e; 1][:, b]^T      (TALL_IS_Z: the wide side is Z)
//   layer 1:  W1x-bar (100 x 786) = sum over e, b of                             z1-bar_e[:, b] [g_e; t_e; 1][:, b]^T      (the wide side is [X; t; 1])
// K = batch columns x evaluations (512 x ~290 per step: 29 GFLOP), both operands come off the tape in fp32.  rnde_wgrad3_kernel multiplies them with
// v_mfma_f32_16x16x4_f32 -- on gfx950 an instruction of the vector ALUs (rnde_x3.h) -- at ~58 % of that unit's peak.  Here every 32-column step of
// both operands is split EXACTLY into three bf16 planes on its way into LDS and the six leading cross products run as v_mfma_f32_16x16x32_bf16: per
// step and wave 168 matrix instructions of 16 cycles where the fp32 form issues 224 of 32, and the splitting (~180 vector instructions per thread and
// step) executes beside them.  Same work decomposition as rnde_wgrad3_kernel: a workgroup of 7 waves owns one half of the wide side (<= 25 tiles of 16
// rows) against the whole narrow side (7 tiles), wave w the wide tiles w, w + 7, w + 14, (w + 21); chunks of steps -> per-chunk slabs, reduced in a
// fixed order afterwards (deterministic).  The operands travel global -> registers (eight scalar loads per (row, 8 columns) unit, coalesced over rows,
// requested one step ahead) -> split -> LDS planes [plane][row][32 k-values + pad] -> b128 fragment reads.
#pragma once
#include "rnde_x3.h"

namespace rnde {

constexpr int kWxRowShorts = 40;                       // bf16 per (plane, row) of a step's image: 32 k-values + 8 of padding (80 bytes: the 16 rows of a fragment read hit 64 different banks)
constexpr int kWxTallRows = 400, kWxNarrowRows = 112;
constexpr int kWxPlaneShortsT = kWxTallRows * kWxRowShorts, kWxPlaneShortsS = kWxNarrowRows * kWxRowShorts;
constexpr size_t kWxLdsBytes = (size_t)3 * (kWxPlaneShortsT + kWxPlaneShortsS) * 2;      // 122,880 bytes

template <bool TALL_IS_Z>
__global__ __launch_bounds__(448) void rnde_wgrad3x_kernel(const EvalDesc* __restrict__ evals, int n_evals, int per_chunk, int M, int Nx, int Bpad, float* __restrict__ slab) {
    constexpr int KC = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short wxs[];
    unsigned short* TP = wxs;                                     // [3][400][40]
    unsigned short* SP = wxs + 3 * kWxPlaneShortsT;               // [3][112][40]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow = lane & 15, kk = lane >> 4;
    const int TR = TALL_IS_Z ? M : Nx + 2, SR = TALL_IS_Z ? Nx + 2 : M;          // rows of the wide / narrow operand, synthetic {t, 1} rows included
    const int TRp = TALL_IS_Z ? M : Nx, SRp = TALL_IS_Z ? Nx : M;                 // rows that exist in memory
    const int TT = (TR + 15) >> 4, T0 = (TT + 1) >> 1;
    const int half = blockIdx.x, chunk = blockIdx.y;
    const int tile_lo = half ? T0 : 0, tile_hi = half ? TT : T0, ntile = tile_hi - tile_lo;     // <= 25
    const int row_lo = 16 * tile_lo, nrow = 16 * ntile;                                           // <= 400
    const int steps_per_eval = (Bpad + KC - 1) / KC;
    const int s_lo = chunk * per_chunk, s_hi = min(n_evals * steps_per_eval, s_lo + per_chunk);
    const int total_steps = max(0, s_hi - s_lo);
    x3f4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 7; ++n) acc[i][n] = (x3f4){0.f, 0.f, 0.f, 0.f};
    const bool has3 = w + 21 < ntile;                      // tiles w, w + 7, w + 14 always exist (host: both halves >= 21 tiles)

    // ---- units: (operand, row, k-octet o < 4) -> eight values (k = 8 o + j) of one row.  Wide units u = tid + 448 jj, jj < 4 (u < 4 nrow <= 1600), narrow
    // units v = tid (4 x 112 = 448: one per thread), so that every load instruction of a wave reads ONE array.  Buffer loads: the per-lane byte offset of a
    // unit is loop invariant, the column of the step comes in through an add, and whatever has no source (padding rows, columns past the batch, units that
    // do not exist) is out of the descriptor's range and loads 0 -- no branches, no 64-bit addresses (the first build of this kernel spent 40 registers on
    // those and spilled).  kind: 0 = nothing, 1 = rows in memory, 2 = the synthetic t row, 3 = the synthetic 1 row, 4 = zeros.
    constexpr unsigned kNoSrc = 0x7FFFFF00u;
    int u_kind[5], u_dst[5];
    unsigned u_voff[5];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int u = tid + 448 * jj;
        int kind = 0, dst = 0; unsigned voff = kNoSrc;
        if (u < 4 * nrow) {
            const int o = u / nrow, row = u - o * nrow, gr = row_lo + row;
            kind = gr < TRp ? 1 : (!TALL_IS_Z && gr == TRp ? 2 : (!TALL_IS_Z && gr == TRp + 1 ? 3 : 4));
            if (kind == 1) voff = 4u * (unsigned)(gr + 8 * o * TRp);      // byte offset from the step's first column (column stride TRp)
            dst = row * kWxRowShorts + 8 * o;
        }
        u_kind[jj] = kind; u_voff[jj] = voff; u_dst[jj] = dst;
    }
    {
        const int o = tid / kWxNarrowRows, row = tid - o * kWxNarrowRows;
        const int kind = row < SRp ? 1 : (TALL_IS_Z && row == SRp ? 2 : (TALL_IS_Z && row == SRp + 1 ? 3 : 4));
        u_kind[4] = kind; u_voff[4] = kind == 1 ? 4u * (unsigned)(row + 8 * o * SRp) : kNoSrc;
        u_dst[4] = 3 * kWxPlaneShortsT + row * kWxRowShorts + 8 * o;      // (into SP)
    }
    int e_nx = s_lo / steps_per_eval, cs_nx = s_lo - e_nx * steps_per_eval;   // (evaluation, 32-column step in it) of the next step to fetch
    float stg[5][8];
    float stg_t = 0.f; int stg_ncols = 0;
    auto fetch = [&]() {                                   // the next step's values -> registers (requests only; first use is in `spill`)
        const int e = e_nx, c0 = cs_nx * KC;
        if (++cs_nx == steps_per_eval) { cs_nx = 0; ++e_nx; }
        stg_t = evals[e].t;
        stg_ncols = min(KC, Bpad - c0);
        // descriptors over the WHOLE arrays (rows x Bpad floats): a column past Bpad is out of range
        __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(TALL_IS_Z ? evals[e].Z : evals[e].X), 0, 4 * TRp * Bpad, 0x00020000);
        __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(TALL_IS_Z ? evals[e].X : evals[e].Z), 0, 4 * SRp * Bpad, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int cs = jj < 4 ? TRp : SRp;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned off = u_voff[jj] + 4u * (unsigned)((c0 + j) * cs);
                stg[jj][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(jj < 4 ? rsT : rsS, (int)off, 0, 2));      // aux 2 = nt: read once
            }
        }
    };
    auto spill = [&]() {                                   // split the fetched values and write the three planes of the step's image
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int kind = u_kind[jj];
            if (kind == 0) continue;
            const int o8 = u_dst[jj] % kWxRowShorts;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = kind == 1 ? stg[jj][j] : ((kind == 2 || kind == 3) && o8 + j < stg_ncols ? (kind == 2 ? stg_t : 1.f) : 0.f);
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x3_split2(v[2 * j], v[2 * j + 1], hi[j], mid[j], lo[j]);
            unsigned short* d = wxs + u_dst[jj];
            const int ps = jj < 4 ? kWxPlaneShortsT : kWxPlaneShortsS;
            *(x3u4*)d = (x3u4){hi[0], hi[1], hi[2], hi[3]};
            *(x3u4*)(d + ps) = (x3u4){mid[0], mid[1], mid[2], mid[3]};
            *(x3u4*)(d + 2 * ps) = (x3u4){lo[0], lo[1], lo[2], lo[3]};
        }
    };
    auto fragT = [&](int tile, int pl) { return *(const x3u4*)(TP + pl * kWxPlaneShortsT + (16 * tile + mrow) * kWxRowShorts + 8 * kk); };
    auto fragS = [&](int tile, int pl) { return *(const x3u4*)(SP + pl * kWxPlaneShortsS + (16 * tile + mrow) * kWxRowShorts + 8 * kk); };

    if (total_steps > 0) { fetch(); spill(); }
    __syncthreads();
    for (int step = 0; step < total_steps; ++step) {
        if (step + 1 < total_steps) fetch();               // in flight under this step's matrix instructions
        // two passes over the narrow side, each with the A fragments of TWO of the wave's wide tiles in registers (all four: 48 registers more than the
        // 256 a wave of this workgroup may hold -- the first build spilled 59 dwords); the B fragments are read twice: 54 KB per wave and step, half the
        // LDS bandwidth the matrix instructions leave time for.  Six leading terms, smallest first, into the tile's ONE accumulator (6 roundings per 32
        // k-values where the fp32 chain has 32), term by term over the pass's two tiles: two instructions on one accumulator are two apart.
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            x3u4 a[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ii = 2 * ps + i, tile = (ii < 3 || has3) ? w + 7 * ii : w;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = fragT(tile, pl);
            }
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting the B fragments of all seven tiles in front of the first matrix instruction: 84 registers)
                const x3u4 bh = fragS(n, 0), bm = fragS(n, 1), bl = fragS(n, 2);
#define WX_TERM(PA, BV) _Pragma("unroll") for (int i = 0; i < 2; ++i) if (2 * ps + i < 3 || has3) acc[2 * ps + i][n] = x3_mfma(a[i][PA], BV, acc[2 * ps + i][n]);
                WX_TERM(2, bh) WX_TERM(0, bl) WX_TERM(1, bm) WX_TERM(1, bh) WX_TERM(0, bm) WX_TERM(0, bh)
#undef WX_TERM
            }
        }
        __syncthreads();                                    // every wave has read this step's image
        if (step + 1 < total_steps) spill();
        __syncthreads();
    }
    // D register q of lane l = C[wide row 16 T + 4 (l >> 4) + q][narrow row 16 n + (l & 15)]
    float* out = slab + (size_t)chunk * M * (Nx + 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (w + 7 * i < ntile) {
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                const int sr = 16 * n + mrow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tr = row_lo + 16 * (w + 7 * i) + 4 * kk + q;
                    if (tr < TR && sr < SR) out[TALL_IS_Z ? (size_t)sr * M + tr : (size_t)tr * M + sr] = acc[i][n][q];
                }
            }
        }
    }
}

// ---- the QUARTER form (round 6, second build): the half form above is single-buffered -- 120 KB of image per workgroup -- so the splitting of step s + 1 (vector
// instructions, LDS writes) and the matrix instructions of step s take turns between two barriers: 12.5 k cycles per step where the matrix instructions need
// 5.4 k (rocprof: 30-33 % matrix-pipe busy).  Here a workgroup owns a QUARTER of the wide side (<= 13 tiles of 16 rows) against the whole narrow side: the
// image is 76.8 KB, TWO of them fit, step s + 1 is split and written while step s is multiplied (one barrier per step), the vector work of a wave sits between
// its own matrix instructions and beside those of its SIMD partner.  Wave w owns wide tiles w and w + 7 (one pass over the narrow side, both tiles' A fragments
// in registers); the narrow operand is split by four workgroups instead of two (+22 % vector work and L2 reads), the four parts of a chunk sit on ONE XCD
// (linear workgroup id -> (xcd, chunk, part) below) so that its L2 serves three of those four reads.  Same six terms in the same order per 32 columns, same
// per-chunk slabs and fixed-order reduction as the half form.
#ifndef RNDE_WX4_SB
#define RNDE_WX4_SB 1
#endif
#ifndef RNDE_WX4_ABL      // timing ablations (wrong results): 1 no splitting, 2 no splitting and no loads, 3 B fragments read once per step, 4 matrix instructions only, 5 split without its LDS writes, 6 LDS writes without the split
#define RNDE_WX4_ABL 0
#endif
constexpr int kWx4TallRows = 208;
constexpr int kWx4PlaneT = kWx4TallRows * kWxRowShorts, kWx4PlaneS = kWxNarrowRows * kWxRowShorts;
constexpr int kWx4ImageShorts = 3 * (kWx4PlaneT + kWx4PlaneS);      // 38,400 bf16
constexpr size_t kWx4LdsBytes = (size_t)2 * kWx4ImageShorts * 2;    // 153,600 bytes: two images

template <bool TALL_IS_Z>
__global__ __launch_bounds__(448) void rnde_wgrad4x_kernel(const EvalDesc* __restrict__ evals, int n_evals, int per_chunk, int n_chunks, int M, int Nx, int Bpad,
                                                           float* __restrict__ slab) {
    constexpr int KC = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short wxs[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow = lane & 15, kk = lane >> 4;
    // linear id -> XCD (round-robin dispatch: id % 8), then chunk-major inside the XCD: the four parts of a chunk are neighbours on one XCD
    const int lin = blockIdx.x, xcd = lin & 7, within = lin >> 3;
    const int chunk = (within >> 2) * 8 + xcd, part = within & 3;
    if (chunk >= n_chunks) return;
    const int TR = TALL_IS_Z ? M : Nx + 2, SR = TALL_IS_Z ? Nx + 2 : M;          // rows of the wide / narrow operand, synthetic {t, 1} rows included
    const int TRp = TALL_IS_Z ? M : Nx, SRp = TALL_IS_Z ? Nx : M;                 // rows that exist in memory
    const int TT = (TR + 15) >> 4, base = TT >> 2, rem = TT & 3;
    const int ntile = base + (part < rem ? 1 : 0), tile_lo = part * base + (part < rem ? part : rem);      // <= 13 (host: TT <= 52)
    const int row_lo = 16 * tile_lo, nrow = 16 * ntile;                                                      // <= 208
    const int steps_per_eval = (Bpad + KC - 1) / KC;
    const int s_lo = chunk * per_chunk, s_hi = min(n_evals * steps_per_eval, s_lo + per_chunk);
    const int total_steps = max(0, s_hi - s_lo);
    x3f4 acc[2][7];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 7; ++n) acc[i][n] = (x3f4){0.f, 0.f, 0.f, 0.f};
    const bool has2 = w + 7 < ntile;                       // tile w always exists (host: every part >= 7 tiles)
    const int tile1 = has2 ? w + 7 : w;                    // (a wave without a second tile multiplies its first one again and drops the result: no branch per matrix instruction)

    // ---- units: FOUR consecutive rows x EIGHT consecutive columns of one operand per thread, i.e. eight 16-byte loads (one per column: the operands are
    // rows-contiguous) where the half form issues 24 four-byte ones -- a wave's load instruction costs the texture path the same 16 cycles whatever its width.
    // Waves 0..3 split the wide operand (unit = tid mod nrow: quad tid >> 2 of the part's rows, column octet tid & 3), waves 4..6 the narrow one (unit =
    // (tid - 256) mod 112); which operand is WAVE-uniform (one buffer descriptor per wave, scalar selects), threads past the last unit repeat an earlier one
    // (same source, destination and values): the loop body has no branch of either kind.  Lane -> (quad, octet) makes the 16 lanes of a b128 LDS write cover
    // the 64 banks exactly once (row stride 80 bytes: bank 20 r; 4 rows of a quad apart 16 banks, octets 4 banks).  Rows past the operand's end are the
    // synthetic {t, 1, 0, 0} quad or zeros: their loads are out of the descriptor's range (they return 0) and the values come from two per-thread constants.
    constexpr unsigned kNoSrc = 0x7FFFFF00u;
    const bool wide_wave = w < 4;
    const int Rp = wide_wave ? TRp : SRp;                                        // rows of this wave's operand in memory = its column stride
    const int nunit = wide_wave ? nrow : kWxNarrowRows;                          // 4 octets x (rows / 4) quads
    const int uid = (wide_wave ? tid : tid - 256) % nunit;
    const int uq = uid >> 2, uo = uid & 3;
    const int u_gr = (wide_wave ? row_lo : 0) + 4 * uq;
    const bool u_mem = u_gr < Rp;
    const bool u_syn = u_gr == Rp && (wide_wave ? !TALL_IS_Z : TALL_IS_Z);     // the quad {t, 1, 0, 0}
    const float syn_f = u_syn ? 1.f : 0.f;
    const unsigned u_voff = u_mem ? 4u * (unsigned)(u_gr + 8 * uo * Rp) : kNoSrc;
    const int u_dst = (wide_wave ? 0 : 3 * kWx4PlaneT) + 4 * uq * kWxRowShorts + 8 * uo;
    const int u_ps = wide_wave ? kWx4PlaneT : kWx4PlaneS;
    // (evaluation, 32-column step in it) of the next step to fetch; past the chunk's last step the last one is fetched again (harmless, never written to LDS)
    int e_nx = s_lo / steps_per_eval, cs_nx = s_lo - e_nx * steps_per_eval, left = total_steps;
    // two register sets: the set a step's splitting reads was requested a step and a half earlier (the operands are read once, cold from HBM: with ONE set,
    // requested behind the splitting and consumed at the head of the next step, every step waited for memory)
    x3f4 stg[2][8];
    float stg_t[2] = {0.f, 0.f}; int stg_ncols[2] = {0, 0};
    auto fetch = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int e = e_nx, c0 = cs_nx * KC;
        {   // (selects, not branches: the loop body stays one basic block)
            const bool adv = --left > 0, wrap = cs_nx + 1 == steps_per_eval;
            e_nx = (adv && wrap) ? e_nx + 1 : e_nx;
            cs_nx = adv ? (wrap ? 0 : cs_nx + 1) : cs_nx;
        }
        stg_t[SET] = evals[e].t;
        stg_ncols[SET] = min(KC, Bpad - c0);
        const float* src = wide_wave == TALL_IS_Z ? evals[e].Z : evals[e].X;      // wide is Z for layer 2 (TALL_IS_Z), narrow is Z for layer 1
        // descriptor over the WHOLE array (rows x Bpad floats): a column past Bpad is out of range and loads 0
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4 * Rp * Bpad, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            stg[SET][j] = __builtin_bit_cast(x3f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)u_voff, 4 * (c0 + j) * Rp, 0));      // (column through the scalar offset: no vector address arithmetic)
    };
    unsigned abl_sink = 0u;
    auto spill_row = [&](unsigned short* img, int i, auto set_c) {      // row i < 4 of the thread's quad: eight values -> three 16-byte plane entries
        constexpr int SET = decltype(set_c)::value;
        // rows that are not in memory were loaded out of range (= 0): the synthetic quad's {t, 1} come in by ONE addition per value of its two rows (0 for everybody
        // else: x + 0 is x), not by a select per value of all four -- 17 vector instructions per step instead of 48.  No test on the column: past the batch's padded
        // width the OTHER operand (always rows in memory) is out of range and 0, so what stands in a synthetic row there multiplies nothing.
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = i < 2 ? stg[SET][j][i] + (i == 0 ? syn_f * stg_t[SET] : syn_f) : stg[SET][j][i];
        unsigned hi[4], mid[4], lo[4];
#if RNDE_WX4_ABL == 6      // (timing ablation: the LDS writes without the split)
#pragma unroll
        for (int j = 0; j < 4; ++j) { hi[j] = __float_as_uint(v[2 * j]); mid[j] = __float_as_uint(v[2 * j + 1]); lo[j] = hi[j]; }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) x3_split2(v[2 * j], v[2 * j + 1], hi[j], mid[j], lo[j]);
#endif
        unsigned short* d = img + u_dst + i * kWxRowShorts;
#if RNDE_WX4_ABL == 5      // (timing ablation: the split without its LDS writes -- the planes are folded into a register the epilogue stores)
        abl_sink ^= hi[0] ^ hi[1] ^ hi[2] ^ hi[3] ^ mid[0] ^ mid[1] ^ mid[2] ^ mid[3] ^ lo[0] ^ lo[1] ^ lo[2] ^ lo[3];
        (void)d;
#else
        *(x3u4*)d = (x3u4){hi[0], hi[1], hi[2], hi[3]};
        *(x3u4*)(d + u_ps) = (x3u4){mid[0], mid[1], mid[2], mid[3]};
        *(x3u4*)(d + 2 * u_ps) = (x3u4){lo[0], lo[1], lo[2], lo[3]};
#endif
    };
    unsigned short* const img0 = wxs;
    unsigned short* const img1 = wxs + kWx4ImageShorts;
    if (total_steps == 0) return;      // (workgroup-uniform; the slab of such a chunk is never read: the host counts chunks from the steps)
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    fetch(S0{});                                          // step 0
#pragma unroll
    for (int i = 0; i < 4; ++i) spill_row(img0, i, S0{});
    fetch(S0{});                                          // step 1 -> set 0 (split during step 0)
    fetch(S1{});                                          // step 2 -> set 1 (split during step 1)
    __syncthreads();
    // one step: the matrix instructions of image `cur`; MORE: the next step's image is written from register set SET -- one row of the thread's quad in each
    // of the blocks n = 0..3, in ONE basic block with that block's twelve matrix instructions -- and the set is refilled behind them with the step after next
    auto one_step = [&](const unsigned short* cur, unsigned short* nxt, auto more_c, auto set_c) {
        constexpr bool MORE = decltype(more_c)::value;
        const unsigned short* SPc = cur + 3 * kWx4PlaneT;
        auto fragT = [&](int tile, int pl) { return *(const x3u4*)(cur + pl * kWx4PlaneT + (16 * tile + mrow) * kWxRowShorts + 8 * kk); };
        auto fragS = [&](int tile, int pl) { return *(const x3u4*)(SPc + pl * kWx4PlaneS + (16 * tile + mrow) * kWxRowShorts + 8 * kk); };
        x3u4 a[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { a[0][pl] = fragT(w, pl); a[1][pl] = fragT(tile1, pl); }
        x3u4 bh = fragS(0, 0), bm = fragS(0, 1), bl = fragS(0, 2);
#pragma unroll
        for (int n = 0; n < 7; ++n) {
            x3u4 nh = bh, nm = bm, nl = bl;
#if RNDE_WX4_ABL != 3 && RNDE_WX4_ABL != 4
            if (n < 6) { nh = fragS(n + 1, 0); nm = fragS(n + 1, 1); nl = fragS(n + 1, 2); }
#endif
#if RNDE_WX4_ABL != 1 && RNDE_WX4_ABL != 2 && RNDE_WX4_ABL != 4
            if (MORE && n < 4) spill_row(nxt, n, set_c);
#endif
#if RNDE_WX4_ABL != 2 && RNDE_WX4_ABL != 4
            if (MORE && n == 4) fetch(set_c);
#endif
#define WX4_TERM(PA, BV) { acc[0][n] = x3_mfma(a[0][PA], BV, acc[0][n]); acc[1][n] = x3_mfma(a[1][PA], BV, acc[1][n]); }
            WX4_TERM(2, bh) WX4_TERM(0, bl) WX4_TERM(1, bm) WX4_TERM(1, bh) WX4_TERM(0, bm) WX4_TERM(0, bh)
#undef WX4_TERM
            bh = nh; bm = nm; bl = nl;
            // (block boundary for the scheduler: a row's ~70 vector instructions stay with THESE twelve matrix instructions -- left alone the compiler front-loads
            //  the splitting of all four rows and issues the last ~40 matrix instructions back to back, and the two waves of a SIMD, released by the same barrier,
            //  do so in step: vector phase against vector phase, matrix phase against matrix phase)
#if RNDE_WX4_SB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    int step = 0;
    for (; step + 2 < total_steps; step += 2) {
        one_step(img0, img1, std::true_type{}, S0{});
        __syncthreads();      // this step's image has been read by every wave, the next one is complete
        one_step(img1, img0, std::true_type{}, S1{});
        __syncthreads();
    }
    if (total_steps - step == 2) {
        one_step(img0, img1, std::true_type{}, S0{});
        __syncthreads();
        one_step(img1, nullptr, std::false_type{}, S1{});
    } else one_step(img0, nullptr, std::false_type{}, S0{});
#if RNDE_WX4_ABL == 5
    if (abl_sink == 0x12345678u) slab[0] = 1.f;
#endif
    float* out = slab + (size_t)chunk * M * (Nx + 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (w + 7 * i < ntile) {
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                const int sr = 16 * n + mrow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tr = row_lo + 16 * (w + 7 * i) + 4 * kk + q;
                    if (tr < TR && sr < SR) out[TALL_IS_Z ? (size_t)sr * M + tr : (size_t)tr * M + sr] = acc[i][n][q];
                }
            }
        }
    }
}

}  // namespace rnde
