// rnde_stage_persist.h -- the stage engine's attempted step as ONE launch.
//
// rnde_stage_kernel (rnde_stage.h) crosses a kernel boundary after every Runge-Kutta stage only to hand the split-K
// partials of layer 1 (the "slab") from the R row blocks of a column tile to each other.  Each boundary costs the launch
// ramp plus a chain of cold, dependent loads (controller state, weights, state arrays): ~9 us per stage, of which ~3 us is
// arithmetic.  Here the 7 stages of an attempt run in one kernel:
//   * weights (this block's 1/R slice, 64 VGPRs per lane) and the attempt's own k_1..k_6 / uprev rows stay in registers;
//   * the slab hand-off happens inside the kernel between the R workgroups of a column tile.  They have the same
//     blockIdx % 8, hence sit on the same XCD (round-robin dispatch; verified at run time from HW_REG_XCC_ID) and share its
//     L2.  A flag-based protocol (stores, release, barrier, flag; poll, acquire, barrier, loads) costs 3.1 us per
//     hand-off (tools/micro/cluster_sync.hip: four dependent L2 round trips); the tagged entries below carry validity in
//     the data itself and need about half of that.  No L2 write-back is involved in either, which is what makes an
//     agent-scope release slow on a multi-XCD part;
//   * the kernel boundary that remains (one per attempt) is the one the algorithm needs: the global error norm.
// Arithmetic, association order and tape layout are exactly those of rnde_stage_kernel, so results are bit-identical
// (tests/test_gpu_forward.py::test_persistent_attempt_is_bit_identical).
// Every spin is bounded: on time-out (workgroups not co-resident, or a placement that is not what was assumed) the
// kernel raises `abort_flag`, all workgroups leave, and the host falls back to the multi-launch kernels for good.
#pragma once
#include "rnde_stage.h"
#include "rnde_x3.h"

#include <type_traits>

namespace rnde {

struct PersistSync {
    float* tslab;           // hand-off slabs [3][C][R][HT][64] f32x4 (see slab_put / slab_poll_sum)
    unsigned* abort_flag;   // [0] a hand-off timed out
    unsigned* xcc;          // [grid] XCC id of each workgroup (written every launch, checked by the host)
    int max_spins;          // bound of every polling loop (kPersistMaxSpins; RNDE_PERSIST_SPINS overrides it: the fallback test uses 0)
};

constexpr int kPersistMaxSpins = 2000000;   // ~1 s: a hand-off only gives up when its partners are truly not running (another tenant keeping CUs busy merely delays them)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- slab hand-off: the data is its own validity ---------------------------------------------------------------------------
// A slab tile (one f32x4 of layer-1 partials per lane) travels as ONE 16-byte entry per lane.  An entry that has not been
// written yet holds the bit pattern kSlabEmpty in all four words (a NaN no arithmetic here produces); the consumer's polling
// load is the data load, and the data is valid when none of the four words is kSlabEmpty (seen from the SUM of the entries, which
// is what the consumer wants anyway: no NaN in it, nothing was empty) -- no flag, no tag words, no release wait, no second load.  Polling volume is what a hand-off costs (tools/micro/cluster_poll_size.hip: 1.69 us per exchange with
// two tagged entries per lane, the previous form of this protocol, 0.99 us with one), so the entry carries payload only.
//
// Who empties an entry again: its producer, and it can tell when that is safe.  Every launch that exchanges at all performs
// exactly 6 exchanges (ex = 1..6, forward and reverse kernels alike), exchange ex uses buffer ex % 3.  When a workgroup's poll of
// exchange ex succeeds, every row block of its column tile has produced ex, hence consumed ex - 1: the workgroup empties ITS
// entries of buffer (ex - 1) % 3 (for ex = 1 that is buffer 0 = exchange 6 of the previous launch, which the kernel boundary has
// long completed).  It waits for those stores to be acknowledged (s_waitcnt vmcnt(0), placed a whole phase later, where it is free)
// before its next put, so whoever sees that put -- and only such a workgroup can get to polling exchange ex + 2 in the buffer just
// emptied -- also sees the emptied entries, never the stale ones.  6 = 0 (mod 3), so every launch starts in the same state
// (buffers 1, 2 empty, buffer 0 holding the last exchange), and launches that exit at once (a finished solve) change nothing.
// The host fills the slabs with kSlabEmpty at creation and whenever the batch width (the tile indexing) changes.
// (A back-off between polls, s_sleep 4 / 16, was measured and makes the attempt 1-3 % slower.)
constexpr unsigned kSlabEmpty = 0xFFFFFFFFu;
__device__ __forceinline__ int slab_buf(unsigned ex) { return (int)(ex % 3u); }
__device__ __forceinline__ void slab_put(float* tslab, size_t tile_index, int lane, const f32x4& v) {
    // a value that happens to carry the empty pattern (only a NaN with every payload bit set can) travels as 0xFFFFFFFE, NaN all the same
    u32x4 b = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = b[i] < 0xFFFFFFFEu ? b[i] : 0xFFFFFFFEu;
    ((u32x4*)tslab + tile_index * 64)[lane] = b;
}
__device__ __forceinline__ void slab_clear(float* tslab, size_t tile_index, int lane) {
    const float e = __builtin_bit_cast(float, kSlabEmpty);
    ((f32x4*)tslab + tile_index * 64)[lane] = (f32x4){e, e, e, e};
}
// all earlier vector-memory operations of this wave (the slab_clear stores) acknowledged; gfx9 encoding of vmcnt(0) alone
__device__ __forceinline__ void slab_clears_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// Sum over the R row blocks (fixed order r = 0..R-1, as the multi-launch kernels) of tile `ht` of column tile `ct` in buffer
// `buf`; polls until all R entries are written.  Returns false on time-out / abort (per wave; the caller agrees over the
// workgroup at its next barrier).
__device__ __forceinline__ bool slab_poll_sum(const PersistSync& Y, int buf, int C, int R, int HT, int ct, int ht, int lane, f32x4& zs) {
    const float* base = Y.tslab + ((((size_t)buf * C + ct) * R) * HT) * 256;     // R * HT tiles of 64 f32x4 = 256 floats
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    while (true) {
        // compiler barrier: the buffer loads below are plain (non-volatile) reads to the optimiser, which would otherwise be free
        // to hoist them out of the spin loop (tools/micro/cluster_poll_size.hip showed exactly that happening to a loop without
        // the atomic abort-flag load further down)
        __asm__ volatile("" ::: "memory");
        u32x4 e[kSMaxW];
#pragma unroll
        for (int r = 0; r < kSMaxW; ++r)
            if (r < R) e[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r * HT + ht) * 64 + lane) * 16, 0, 16);   // aux 16 = sc1: agent scope, misses L1
        // The sums are formed at once (fixed order r = 0..R-1) and THEY say whether every entry had arrived: the empty pattern is a NaN, so
        // an entry still empty leaves a NaN in its sum -- two unordered-compares where the word-by-word check was 17 instructions per poll.
        // (scalar adds on purpose: the vector form `zs += bitcast(e[r])` over buffer-load results was miscompiled by this
        //  toolchain into v_pk_add_f32 with op_sel_hi:[0,0] in the tagged form of this helper -- two of four sums wrong)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int r = 0; r < kSMaxW; ++r) {
            if (r < R) {
                const f32x4 f = __builtin_bit_cast(f32x4, e[r]);
                s0 += f[0]; s1 += f[1]; s2 += f[2]; s3 += f[3];
            }
        }
        zs = (f32x4){s0, s1, s2, s3};
        const bool clean = !(__builtin_isunordered(s0, s1) || __builtin_isunordered(s2, s3));
        if (__all(clean)) return true;
        // a NaN somewhere: entries not written yet -- or NaN DATA (a diverging solve; slab_put keeps such values off the empty pattern), which
        // must pass: "no word is kSlabEmpty" == "the unsigned maximum of all words is not 0xFFFFFFFF" (v_max3_u32 chain, one compare)
        unsigned m = 0u;
#pragma unroll
        for (int r = 0; r < kSMaxW; ++r)
            if (r < R) { const unsigned a = e[r][0] > e[r][1] ? e[r][0] : e[r][1], b = e[r][2] > e[r][3] ? e[r][2] : e[r][3]; const unsigned c = a > b ? a : b; m = m > c ? m : c; }
        const bool ok = m != kSlabEmpty;
        if (__all(ok)) return true;
        if (++spins > Y.max_spins || ((spins & 63) == 0 && __hip_atomic_load(Y.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(Y.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            zs = (f32x4){0.f, 0.f, 0.f, 0.f};
            return false;
        }
    }
}

// (the ablation builds of this kernel -- no polls / no tanh / no tape / MFMAs alone, profiles/r0N_attempt_ablation.csv -- are made by patching a
//  copy of this file: tools/experiments/attempt_ablation/)
#ifdef RNDE_DIAG
#define PSTAMP(i) do { if (P.dbg_out && wg == 0 && tid == 0) ((unsigned long long*)P.dbg_out)[i] = clock64(); } while (0)
// per-wave stamps of workgroup 0: [64 + ((stage - 1) * 8 + wave) * 8 + k]
#define WSTAMP(st, k) do { if (P.dbg_out && wg == 0 && lane == 0) ((unsigned long long*)P.dbg_out)[64 + (((st) - 1) * 8 + w) * 8 + (k)] = clock64(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#define WSTAMP(st, k) do { } while (0)
#endif

// FIX = 1: the geometry of the headline configuration (D = 784 = 49 row tiles, H = 100 -> 7 hidden tiles, 7 waves, 7 row blocks) as
// compile-time constants: the stages are instruction bound and full of wave-uniform conditions on these numbers.  FIX = 0: any geometry.
// X3 = 1 (with FIX): the Dense-layer products on the matrix cores (rnde_x3.h), the same instruction sequence as rnde_stage_solve_kernel<.., 1> -- the two
// are bit-identical to each other, as their fp32-input-MFMA forms are (tests/test_gpu_x3.py).  x3B / x3D: the split weights (rnde_launch_x3_pack).
template <int ACT2, int FIX, int X3 = 0>
__global__ __launch_bounds__(64 * kSMaxW) void rnde_stage_attempt_kernel(const StageParams Q, const int n, const PersistSync Y, const void* x3B = nullptr, const void* x3D = nullptr) {
    static_assert(!X3 || FIX, "the X3 form exists for the headline geometry only");
    const StepParams& P = Q.F;
    const int gWT = FIX ? 7 : Q.WT, gHT = FIX ? 7 : Q.HT, gK2b = FIX ? 7 : Q.K2b, gMT = FIX ? 49 : Q.MT, gR = FIX ? 7 : Q.R;
    const int gD = FIX ? 784 : P.D, gH = FIX ? 100 : P.H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KH = 16 * gK2b + 4, KG = 16 * gWT + 4;
    float* HL = smem;
    float* GL = HL + (X3 ? kX3ImageFloats : kSCB * KH);
    float* RED = GL + (X3 ? kX3ImageFloats : kSCB * KG);         // [32]; RED[24..31] = per-wave "gave up" flags of the hand-off
    unsigned short* HX = (unsigned short*)HL;      // X3: operand images [plane][column][kX3K] of bf16
    unsigned short* GX = (unsigned short*)GL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (FIX) __builtin_assume(w >= 0 && w < 7);
    // Workgroup -> (row block, column tile): the R row blocks of a column tile must sit on ONE XCD (they talk through its
    // L2).  Dispatch is round-robin, XCD = blockIdx % 8, so a tile's members are given block indices that agree mod 8;
    // the grid is 8 * R * ceil(C / 8) and the surplus workgroups (ct >= C) leave at once.
    const int rb = (blockIdx.x >> 3) % gR, ct = 8 * ((blockIdx.x >> 3) / gR) + (blockIdx.x & 7);
    if (ct >= Q.C) return;
    const int wg = rb * Q.C + ct;        // logical workgroup id (index of the per-workgroup partials, as in rnde_stage_kernel)
    const int col = lane & 15, gcol = ct * kSCB + col;
    const bool colok = gcol < P.B;
    const bool vec = (gD & 3) == 0;
    const bool writer = (wg == 0 && tid == 0);
    const int T = rb * gWT + w;
    const int r0 = 16 * T + 4 * (lane >> 4);
    const bool tile_ok = FIX ? true : T < gMT;      // (FIX: 7 x 7 = 49 row tiles, none missing)
    const RecLayout L{(long long)gD * P.Bpad, (long long)gH * P.Bpad};
    if (tid == 0) Y.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID

    PSTAMP(0);
    // In a solve the prologue used to be a chain of THREE cold round trips (~2 k cycles each): the controller state of attempt n - 1, then
    // (behind its `done` test) that attempt's 224 error partials, then -- at an address the controller's verdict decides -- the state the
    // step starts from.  None of the addresses but the last depends on loaded data, and the last has a most likely value: the record
    // attempt n - 1 wrote (it holds (unew, k7) if that attempt was accepted).  So the partials and that record's two tiles are requested
    // here, IN FRONT of the weight loads (a wave's loads return in order: behind the 64 registers of weights the partials arrived ~3 k cycles
    // later than they could), and the controller only has to wait for whichever arrives last.  (Same values, same arithmetic.)
    float pre_part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 prev_raw[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // the 48 bytes of P.ctl[(n - 1) & 1], untouched until the controller
    if (n > 0) {
        const f32x4* cp = (const f32x4*)&P.ctl[(n - 1) & 1];
        prev_raw[0] = cp[0]; prev_raw[1] = cp[1]; prev_raw[2] = cp[2];
        partials_request(P.errpart + (size_t)((n - 1) & 1) * 3 * P.nwg, lane, pre_part);
    }
    const bool spec = P.tape && n > 0 && !P.forced && tile_ok;
    f32x4 sp_up = {0.f, 0.f, 0.f, 0.f}, sp_k = {0.f, 0.f, 0.f, 0.f};
    if (spec) {
        const float* Rs = P.arena + (long long)(n - 1) * P.rec_stride;
        sp_up = ld4(Rs + L.unew() + (size_t)gcol * gD, r0, gD, true, vec);
        sp_k = ld4(Rs + L.k(7) + (size_t)gcol * gD, r0, gD, true, vec);
    }
    // ---- this block's weight slice and the layer-1 bias / time column of this wave's hidden tile: loaded once ----
    // (the small loads first: behind the 14 weight loads their addresses were built in registers of loads in flight -- three more waits)
    float w1t_own[4] = {0.f, 0.f, 0.f, 0.f}, b1_own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        if (hr < gH) { w1t_own[i] = Q.p[(size_t)gH * gD + hr]; b1_own[i] = Q.p[(size_t)gH * (gD + 1) + hr]; }
    }
    // (both lane addresses exist before the first load is issued -- pinned by the empty asm: the register allocator otherwise builds the
    //  second one in registers that are the destination of a load in flight, which costs a full wait for the first seven loads)
    typedef const __attribute__((address_space(1))) f32x4* gw4;     // (through the asm the compiler no longer knows the pointers are global)
    unsigned long long aB = (unsigned long long)(Q.pwB + ((size_t)T * gK2b) * 64 + lane);
    unsigned long long aD = (unsigned long long)(Q.pwD + ((size_t)w * gMT + rb * gWT) * 64 + lane);
    unsigned long long aB4 = aB + 4 * 1024, aD4 = aD + 4 * 1024;       // (the offset field of a load reaches 4095 bytes: k-blocks 4.. need their own base)
    asm volatile("" : "+v"(aB), "+v"(aD), "+v"(aB4), "+v"(aD4));
    f32x4 wB[X3 ? 1 : kSMaxHT], wD[X3 ? 1 : kSMaxW];
    x3u4 xB[X3 ? 4 : 1][3], xD[X3 ? 4 : 1][3];
    unsigned long long xb_addr[3] = {0ull, 0ull, 0ull};
    if constexpr (X3) {
        typedef const __attribute__((address_space(1))) x3u4* gx4;
        unsigned long long bD = (unsigned long long)((const x3u4*)x3D + ((size_t)(w * gR + rb) * 4 * 3) * 64 + lane);
        unsigned long long bB = (unsigned long long)((const x3u4*)x3B + ((size_t)T * 4 * 3) * 64 + lane);
        unsigned long long bD1 = bD + 4 * 1024, bD2 = bD + 8 * 1024, bB1 = bB + 4 * 1024, bB2 = bB + 8 * 1024;
        asm volatile("" : "+v"(bD), "+v"(bB), "+v"(bD1), "+v"(bD2), "+v"(bB1), "+v"(bB2));
#pragma unroll
        for (int f = 0; f < 12; ++f) xD[f / 3][f % 3] = f < 4 ? ((gx4)bD)[(size_t)f * 64] : (f < 8 ? ((gx4)bD1)[(size_t)(f - 4) * 64] : ((gx4)bD2)[(size_t)(f - 8) * 64]);
        xb_addr[0] = bB; xb_addr[1] = bB1; xb_addr[2] = bB2;      // (xB: first multiplied in stage 1's phase B -- requested behind the controller and the state the step starts from, see there)
        // k-values 112 .. 135 of every (plane, column) row are written by nobody: zeroed (ONLY those: no barrier between this loop and START's x3_store4)
        for (int i = tid; i < 2 * 3 * 16 * 12; i += 64 * 7) ((unsigned*)HL)[(i / 12) * (kX3K / 2) + 56 + i % 12] = 0u;
    } else {
    // (wD first: the prologue's own phase D needs it; wB is not multiplied before phase B of the first stage and streams in behind)
#pragma unroll
    for (int kb = 0; kb < kSMaxW; ++kb)
        if (kb < gWT && w < gHT && rb * gWT + kb < gMT) wD[kb] = kb < 4 ? ((gw4)aD)[(size_t)kb * 64] : ((gw4)aD4)[(size_t)(kb - 4) * 64];
#pragma unroll
    for (int kb = 0; kb < kSMaxHT; ++kb)
        if (kb < gK2b && tile_ok) {
            if (FIX && kb == 6) {   // k-steps 104.. of the last block multiply zeros and are left out (phase B): their half of the float4 would be dead
                typedef const __attribute__((address_space(1))) f32x2* gw2;      // registers the moment it is requested, and the allocator's reuse
                const f32x2 lo = *(gw2)(aB4 + 2 * 1024);                       // of them a wait for every load issued before it
                wB[kb] = (f32x4){lo.x, lo.y, 0.f, 0.f};
            } else wB[kb] = kb < 4 ? ((gw4)aB)[(size_t)kb * 64] : ((gw4)aB4)[(size_t)(kb - 4) * 64];
        }
    }
    const float* W1t = Q.p + (size_t)gH * gD;
    const float* b1 = Q.p + (size_t)gH * (gD + 1);

    // ---- controller (identical to SM_START) ----
    // (first USE of the pre-loaded state: the empty asm keeps the compiler from unpacking it -- and waiting for it -- up where it was requested)
    PSTAMP(35);
    asm volatile("" : "+v"(prev_raw[0]), "+v"(prev_raw[1]), "+v"(prev_raw[2]));
    PSTAMP(36);
    static_assert(sizeof(StepState) == 48, "StepState is read as three 16-byte words");
    StepState prev_state;
    __builtin_memcpy(&prev_state, prev_raw, sizeof(StepState));
    prev_state.live = __builtin_amdgcn_readfirstlane(prev_state.live); prev_state.done = __builtin_amdgcn_readfirstlane(prev_state.done);
    const StepState S = advance_state_t<true>(P, n, lane, writer, &P.ctl[n & 1], pre_part, prev_state);
    if (P.nsave > 0) {
        const int lo = (n == 0) ? 0 : P.ctl[(n - 1) & 1].next_save, hi = S.next_save;
        if (hi > lo && tile_ok) {
            if (n == 0) {
                st_tile(P.sv_out + (size_t)gcol * P.nsave * gD, r0, gD, colok, vec, ld_tile(P.x + (size_t)gcol * gD, r0, gD, colok, P.xvec != 0));
            } else {
                const StepState pv = P.ctl[(n - 1) & 1];
                const float dtp_ = (P.t1 - pv.t < pv.dtp) ? (P.t1 - pv.t) : pv.dtp;
                const float* Rp = P.arena + (long long)S.live * P.rec_stride;
                dense_points(P, L, Rp, pv.t, dtp_, S.t, lo, hi, (size_t)gcol * gD, gcol, r0, colok, vec);
            }
        }
    }
    PSTAMP(1);
    if (S.done) return;
    const float t = S.t, dt = (!P.forced && (P.t1 - S.t < S.dtp)) ? (P.t1 - S.t) : S.dtp;
    const int live = S.live;
    const int rec = P.tape ? n + P.rec_shift : (live == 0 ? 1 : 0);
    float* R = P.arena + (long long)rec * P.rec_stride;
    const float* upsrc = P.x; const float* k1p = P.f0; bool upok = colok, upvec = P.xvec != 0;
    if (live >= 0) { const float* Rl = P.arena + (long long)live * P.rec_stride; upsrc = Rl + L.unew(); k1p = Rl + L.k(7); upok = true; upvec = vec; }
    const size_t co = (size_t)gcol * gD;
    f32x4 c_up = {0.f, 0.f, 0.f, 0.f}, c_un = {0.f, 0.f, 0.f, 0.f}, c_k[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) c_k[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (spec && live == n - 1) { c_up = sp_up; c_k[0] = sp_k; }      // (the step starts from what attempt n - 1 wrote: already here)
    else if (tile_ok) { c_up = ld4(upsrc + co, r0, gD, upok, upvec); c_k[0] = ld4(k1p + co, r0, gD, true, vec); }
    if constexpr (X3) {
        // the weight fragments of phase B at the END of the prologue's request queue (a wave's loads return in order; round 6, rnde_bstage_persist.h: the same move
        // took 0.9 us off a reversed attempt): START's own phase D waits for xD, the controller's inputs and the state only
        typedef const __attribute__((address_space(1))) x3u4* gx4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 12; ++f) xB[f / 3][f % 3] = ((gx4)xb_addr[f >> 2])[(size_t)(f & 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }

    // Loop-invariant addressing of this lane's four rows of its own hidden tile (phase A) and row tile (phase D): the stages are
    // instruction bound between the hand-offs (7 waves share 4 SIMDs), so nothing that does not change is recomputed per stage.
    int own_hl[4], own_kind[4], own_gl[4];        // LDS offsets (-1: no slot); kind 0 = hidden unit, 1 = the t row, 2 = the 1 row, 3 = padding
    size_t own_hd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hr = 16 * w + 4 * (lane >> 4) + i;
        own_kind[i] = hr < gH ? 0 : (hr == gH ? 1 : (hr == gH + 1 ? 2 : 3));
        own_hl[i] = hr < 16 * gK2b ? col * KH + kperm(hr) : -1;
        own_hd[i] = (size_t)gcol * gH + hr;
        own_gl[i] = col * KG + kperm(hr);
    }
    // FIX: a lane's four rows of its hidden tile are consecutive hidden units (H = 100 is a multiple of 4) -- one 16-byte tape store --
    // and only wave 6 holds other rows (t, 1, padding): their value is own_c1 * ts + own_c0
    float own_c1[4], own_c0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { own_c1[i] = own_kind[i] == 1 ? 1.f : 0.f; own_c0[i] = own_kind[i] == 2 ? 1.f : 0.f; }
    // (FIX: the hidden activations are computed identically by all seven row blocks of a tile; row block rb tapes hidden tile rb, see rnde_stage_solve.h)
    const bool own_hstore = (FIX ? rb == w : rb == 0) && own_kind[3] == 0;
    if (tid == 0) RED[24] = 0.f;                  // "a wave of this workgroup gave up" (written by any such wave; read after the phase-A barrier)
    // phase D: this row block's layer-1 partial of the stage input v -> slab[par], then publish exchange number `ex`
    auto phase_d = [&](const f32x4& v, unsigned ex) {
        if constexpr (X3) {
            x3_store4(GX, col, 16 * w + 4 * (lane >> 4), v);
            __syncthreads();
            const size_t tile0x = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
            slab_put(Y.tslab, tile0x + w, lane, x3_tile<4>(xD, GX, lane));
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) GL[own_gl[i]] = (FIX || (tile_ok && r0 + i < gD)) ? v[i] : 0.f;
        __syncthreads();
        const size_t tile0 = (((size_t)slab_buf(ex) * Q.C + ct) * gR + rb) * gHT;
        const float* gbp = GL + col * KG + 4 * (lane >> 4);
        f32x4 bg[kSMaxW];
#pragma unroll
        for (int kb = 0; kb < kSMaxW; ++kb) if (kb < gWT) bg[kb] = *(const f32x4*)(gbp + 16 * kb);
        if (w < gHT) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < gWT && rb * gWT + kb < gMT) {
                    acc0 = mfma16(wD[kb][0], bg[kb][0], acc0);
                    acc1 = mfma16(wD[kb][1], bg[kb][1], acc1);
                    acc0 = mfma16(wD[kb][2], bg[kb][2], acc0);
                    acc1 = mfma16(wD[kb][3], bg[kb][3], acc1);
                }
            }
            slab_put(Y.tslab, tile0 + w, lane, acc0 + acc1);
        }
        for (int ht = w + gWT; ht < gHT; ht += gWT) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < kSMaxW; ++kb) {
                if (kb < gWT && rb * gWT + kb < gMT) {
                    const f32x4 a = Q.pwD[((size_t)ht * gMT + rb * gWT + kb) * 64 + lane];
                    acc0 = mfma16(a[0], bg[kb][0], acc0);
                    acc1 = mfma16(a[1], bg[kb][1], acc1);
                    acc0 = mfma16(a[2], bg[kb][2], acc0);
                    acc1 = mfma16(a[3], bg[kb][3], acc1);
                }
            }
            slab_put(Y.tslab, tile0 + ht, lane, acc0 + acc1);
        }
    };

    // ---- SM_START's phase C / D ----
    {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tile_ok) {
            v = fma4(dt, tsA(1, 0) * c_k[0], c_up);
            if (P.tape) st4(R + L.g(2) + co, r0, gD, true, vec, v);
            // (uprev, k1) copies: only the dense output of saveat reads them here (the multi-launch STAGE kernels also do; streaming
            //  tape stores with `nt` were measured: no difference)
            if (P.nsave > 0) { st4(R + L.upc() + co, r0, gD, true, vec, c_up); st4(R + L.k1c() + co, r0, gD, true, vec, c_k[0]); }
        }
        PSTAMP(2);
        phase_d(v, 1u);
        PSTAMP(3);
    }

    float part0 = 0.f, part1 = 0.f, part2 = 0.f;
    bool alive = true;
    // one stage: s = 1..5 -> SM_STAGE, s = 6 -> SM_LAST (zero-based stage index as in rnde_stage_kernel)
    auto stage = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (!alive) return;
        const float ts = fmaf(tsC(s), dt, t);
        float* hdst = R + L.h(s + 1);
        float* kdst = R + L.k(s + 1);
        const int buf = slab_buf((unsigned)s);              // exchange s: put by the previous stage (START for s = 1)
        // ---- phase A: poll this wave's hidden tile of the R row blocks (the polling load is the data load) ----
        bool dead = false;
        f32x4 zs = {0.f, 0.f, 0.f, 0.f};
        if (w < gHT) dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, w, lane, zs);
        PSTAMP(4 + 5 * (s - 1));
        WSTAMP(s, 0);
        // every row block has produced exchange s, hence consumed s - 1: this wave's entries of that buffer can be emptied
        const size_t tprev0 = (((size_t)slab_buf((unsigned)(s + 2)) * Q.C + ct) * gR + rb) * gHT;     // (s - 1) % 3 == (s + 2) % 3
        if constexpr (FIX) {
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = fmaf(w1t_own[i], ts, zs[i]) + b1_own[i];     // (rows that are no hidden unit: coefficients 0, value unused)
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            f32x4 hv = {t01.x, t01.y, t23.x, t23.y};
            if (w == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = own_kind[i] == 0 ? hv[i] : fmaf(own_c1[i], ts, own_c0[i]);
            }
            if (own_hstore) *(f32x4*)(hdst + own_hd[0]) = hv;
            if constexpr (X3) x3_store4(HX, col, 16 * w + 4 * (lane >> 4), hv);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) HL[own_hl[0] + 4 * i] = hv[i];      // kperm: the four rows of a lane sit 4 floats apart
            }
        } else if (w < gHT) {      // this wave's own hidden tile: addressing precomputed (own_*)
            if (!dead) slab_clear(Y.tslab, tprev0 + w, lane);
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = own_kind[i] == 0 ? fmaf(w1t_own[i], ts, zs[i]) + b1_own[i] : 0.f;
            // two tanh per instruction (v_pk_fma_f32): the 4 rows of this lane as two pairs
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            const float th4[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = own_kind[i] == 0 ? th4[i] : (own_kind[i] == 1 ? ts : (own_kind[i] == 2 ? 1.f : 0.f));
                if (rb == 0 && own_kind[i] == 0) hdst[own_hd[i]] = v;
                if (own_hl[i] >= 0) HL[own_hl[i]] = v;
            }
        }
        for (int ht = w + gWT; ht < gHT; ht += gWT) {      // further hidden tiles (more tiles than waves): the general form
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (!dead) dead = !slab_poll_sum(Y, buf, Q.C, gR, gHT, ct, ht, lane, z);
            if (!dead) slab_clear(Y.tslab, tprev0 + ht, lane);
            const int h0 = 16 * ht + 4 * (lane >> 4);
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                pre[i] = (hr < gH) ? fmaf(W1t[hr], ts, z[i]) + b1[hr] : 0.f;
            }
            const f32x2 t01 = tanh_fast2((f32x2){pre[0], pre[1]}), t23 = tanh_fast2((f32x2){pre[2], pre[3]});
            const float th4[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hr = h0 + i;
                float v = 0.f;
                if (hr < gH) {
                    v = th4[i];
                    if (rb == 0) hdst[(size_t)gcol * gH + hr] = v;
                } else if (hr == gH) v = ts;
                else if (hr == gH + 1) v = 1.f;
                if (hr < 16 * gK2b) HL[col * KH + kperm(hr)] = v;
            }
        }
        if (gK2b > gHT) {
            for (int i = tid; i < kSCB * 16 * gK2b; i += blockDim.x) {
                const int c = i / (16 * gK2b), k = i - c * 16 * gK2b;
                if (k >= 16 * gHT) HL[c * KH + kperm(k)] = (k == gH) ? ts : (k == gH + 1 ? 1.f : 0.f);
            }
        }
        if (dead && lane == 0) RED[24] = 1.f;
        WSTAMP(s, 1);
        __syncthreads();
        WSTAMP(s, 2);
        if constexpr (!X3) { if (RED[24] != 0.f) { alive = false; return; } }      // a wave that gave up takes the whole workgroup with it (uniform after the barrier)
        PSTAMP(5 + 5 * (s - 1));
        // ---- phase B ----
        f32x4 kv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            const float gave_up = RED[24];      // (requested in front of the fragments -- x3_tile's first scheduling barrier keeps it there -- and looked at behind the products)
            kv = x3_tile<4>(xB, HX, lane);
            if (gave_up != 0.f) { alive = false; return; }      // (X3: the flag is read with the fragments, not in front of them -- an LDS round trip per stage less on the chain; a workgroup that gives up has multiplied for nothing)
            if (ACT2) {
                const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
            }
        } else if (tile_ok) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* hb = HL + col * KH + 4 * (lane >> 4);
            f32x4 bf[kSMaxHT];
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) if (kb < gK2b) bf[kb] = *(const f32x4*)(hb + 16 * kb);
#pragma unroll
            for (int kb = 0; kb < kSMaxHT; ++kb) {
                if (kb < gK2b) {      // (FIX: the k-steps past row H + 1 multiply zeros and are left out -- 26 MFMAs instead of 28)
                    acc0 = mfma16(wB[kb][0], bf[kb][0], acc0);
                    if (!FIX || 16 * kb + 4 < gH + 2) acc1 = mfma16(wB[kb][1], bf[kb][1], acc1);
                    if (!FIX || 16 * kb + 8 < gH + 2) acc0 = mfma16(wB[kb][2], bf[kb][2], acc0);
                    if (!FIX || 16 * kb + 12 < gH + 2) acc1 = mfma16(wB[kb][3], bf[kb][3], acc1);
                }
            }
            kv = acc0 + acc1;
            if (ACT2) {
                const f32x2 a01 = tanh_fast2((f32x2){kv[0], kv[1]}), a23 = tanh_fast2((f32x2){kv[2], kv[3]});
                kv = (f32x4){a01.x, a01.y, a23.x, a23.y};
            }
            if constexpr (!FIX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) kv[i] = (r0 + i < gD) ? kv[i] : 0.f;
            }
        }
        PSTAMP(6 + 5 * (s - 1));
        WSTAMP(s, 3);
        // ---- phase C ----
        if constexpr (s < 6) {
            slab_clears_done();      // (issued two phases ago: nothing to wait for in practice) before this stage's put, see slab_put
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tile_ok) {
                st4(kdst + co, r0, gD, true, vec, kv);
                f32x4 acc = tsA(s + 1, 0) * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) if (j < s) acc = fma4(tsA(s + 1, j), c_k[j], acc);
                acc = fma4(tsA(s + 1, s), kv, acc);
                v = fma4(dt, acc, c_up);
                if (s == 5) { st4(R + L.unew() + co, r0, gD, true, vec, v); c_un = v; }
                else if (P.tape) st4(R + L.g(s + 2) + co, r0, gD, true, vec, v);
                c_k[s] = kv;
            }
            PSTAMP(7 + 5 * (s - 1));
            WSTAMP(s, 4);
            phase_d(v, (unsigned)(s + 1));
            PSTAMP(8 + 5 * (s - 1));
            WSTAMP(s, 6);
        } else {
            if (tile_ok) {
                st4(kdst + co, r0, gD, true, vec, kv);
                const f32x4 up = c_up, un = c_un;
                f32x4 acc = tsBt(0) * c_k[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc = fma4(tsBt(j), c_k[j], acc);
                acc = fma4(tsBt(6), kv, acc);
                if (colok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ut = dt * acc[i];
                        const float sk = P.abstol + fmaxf(fabsf(up[i]), fabsf(un[i])) * P.reltol;
                        const float r = ut / sk;
                        part0 = add_square_unfused(part0, r);
                    }
                    if (P.reg_kind >= 2) {
                        f32x4 g6 = tsA(5, 0) * c_k[0];
#pragma unroll
                        for (int j = 1; j < 5; ++j) g6 = fma4(tsA(5, j), c_k[j], g6);
                        g6 = fma4(dt, g6, up);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (r0 + i < gD) {
                                const float d1 = kv[i] - c_k[5][i], d2 = un[i] - g6[i];
                                part1 = add_square_unfused(part1, d1); part2 = add_square_unfused(part2, d2);
                            }
                        }
                    }
                }
            }
        }
    };
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{});
    if (!alive) return;
    PSTAMP(34);

    part0 = wave_sum_f(part0); part1 = wave_sum_f(part1); part2 = wave_sum_f(part2);
    if (lane == 0) { RED[w] = part0; RED[8 + w] = part1; RED[16 + w] = part2; }
    __syncthreads();
    if (tid == 0) {
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int i = 0; i < gWT; ++i) { sa += RED[i]; sb += RED[8 + i]; sc += RED[16 + i]; }
        float* ep = P.errpart + (size_t)(n & 1) * 3 * P.nwg;
        ep[wg] = sa; ep[P.nwg + wg] = sb; ep[2 * P.nwg + wg] = sc;
    }
}

}  // namespace rnde
