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

}  // namespace rnde
