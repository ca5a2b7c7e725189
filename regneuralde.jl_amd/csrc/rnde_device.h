// rnde_device.h -- device-side building blocks shared by the forward and reverse step kernels.
//
// Geometry ("column-owner" design):
//   one workgroup (8 waves, 512 threads) owns BT = 4*NG batch columns for a whole Tsit5 attempt.
//   The Dense layers run on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 rank-1 blocks per
//   instruction, arranged here as RG = 16/NG row-groups x NG column-groups, i.e. one instruction
//   updates a (TR = 64/NG rows) x (BT columns) tile with ONE k.  Lane l = 4*b + x, b = block:
//       A operand : W[tile*TR + (l % TR)][k]
//       B operand : X[k][4*(l / TR) + x]
//       D reg i   : out[tile*TR + 4*(b % RG) + i][4*(b / RG) + x]
//   (layout measured on gfx950 with tools/probe_mfma.hip).
//   A lane therefore owns 4 consecutive rows of one column: 16 contiguous bytes in the
//   column-major D x B state arrays, and the SAME (row, col) set for every array -- so all
//   Runge-Kutta linear combinations, the error estimate and the reverse-mode accumulators are
//   pure register arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

// hipFuncSetAttribute (the dynamic-LDS ceiling of a kernel) is a PER-DEVICE setting: a function-local `static bool` would let the second device
// of a process launch without it.  One bit per device, set once the call has succeeded; two threads racing set it twice, which is harmless.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1ull); }
    void done() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace rnde {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

// ---- Tsit5 tableau (Tsitouras 2011), rounded to fp32 at use --------------
__host__ __device__ constexpr float tsA(int s, int j) {  // s, j zero-based: stage s+1 uses k_{j+1}
    constexpr double A[7][7] = {
        {0},
        {0.161},
        {-0.008480655492356989, 0.335480655492357},
        {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
        {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
        {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
        {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0.0}};
    return (float)A[s][j];
}
__host__ __device__ constexpr float tsC(int s) {
    constexpr double C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
    return (float)C[s];
}
__host__ __device__ constexpr float tsBt(int j) {
    constexpr double BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                              -0.1447110071732629,     0.5823571654525552,     -0.45808210592918697,
                              0.015151515151515152};
    return (float)BT[j];
}
// Run-time indexed copies for the rolled stage loops (double literals rounded to fp32 by the compiler,
// identical to (float)A[s][j] above).
// kFwdShift[s][i] = A[s+1+i][s] (0 beyond the tableau): coefficient of k_s in the input of stage s+1+i
__constant__ float kFwdShift[7][6] = {
    {0.161f, -0.008480655492356989f, 2.8971530571054935f, 5.325864828439257f, 5.86145544294642f, 0.09646076681806523f},
    {0.335480655492357f, -6.359448489975075f, -11.748883564062828f, -12.92096931784711f, 0.01f, 0.f},
    {4.3622954328695815f, 7.4955393428898365f, 8.159367898576159f, 0.4798896504144996f, 0.f, 0.f},
    {-0.09249506636175525f, -0.071584973281401f, 1.379008574103742f, 0.f, 0.f, 0.f},
    {-0.028269050394068383f, -3.290069515436081f, 0.f, 0.f, 0.f, 0.f},
    {2.324710524099774f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
};
// kBwdShift[s][i] = A[s][s-1-i] (0 below the tableau): weight of gbar_s in the cotangent of k_{s-1-i}
__constant__ float kBwdShift[7][6] = {
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.161f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.335480655492357f, -0.008480655492356989f, 0.f, 0.f, 0.f, 0.f},
    {4.3622954328695815f, -6.359448489975075f, 2.8971530571054935f, 0.f, 0.f, 0.f},
    {-0.09249506636175525f, 7.4955393428898365f, -11.748883564062828f, 5.325864828439257f, 0.f, 0.f},
    {-0.028269050394068383f, -0.071584973281401f, 8.159367898576159f, -12.92096931784711f, 5.86145544294642f, 0.f},
    {2.324710524099774f, -3.290069515436081f, 1.379008574103742f, 0.4798896504144996f, 0.01f, 0.09646076681806523f},
};
// kTsA[s][j] = a_{s+1,j+1}: run-time indexed copy of tsA (stage engine)
__constant__ float kTsA[7][8] = {
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.161f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {-0.008480655492356989f, 0.335480655492357f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {2.8971530571054935f, -6.359448489975075f, 4.3622954328695815f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {5.325864828439257f, -11.748883564062828f, 7.4955393428898365f, -0.09249506636175525f, 0.f, 0.f, 0.f, 0.f},
    {5.86145544294642f, -12.92096931784711f, 8.159367898576159f, -0.071584973281401f, -0.028269050394068383f, 0.f, 0.f, 0.f},
    {0.09646076681806523f, 0.01f, 0.4798896504144996f, 1.379008574103742f, -3.290069515436081f, 2.324710524099774f, 0.f, 0.f}};
__device__ __forceinline__ float tsA_rt(int s, int j) { return kTsA[s][j]; }
__constant__ float kTsC[8] = {0.f, 0.161f, 0.327f, 0.9f, 0.9800255409045097f, 1.0f, 1.0f, 0.f};
__constant__ float kTsBt[8] = {-0.001780011052225777f, -0.0008164344596567469f, 0.007880878010261995f, -0.1447110071732629f, 0.5823571654525552f, -0.45808210592918697f, 0.015151515151515152f, 0.f};
// Tsit5 dense output u(t + theta*dt) = uprev + dt * sum_i b_i(theta) k_i  
__host__ __device__ inline void dense_weights(float th, float (&b)[7]) {
    const float t2 = th * th;
    b[0] = -1.0530884977290216f * th * (th - 1.3299890189751412f) * (t2 - 1.4364028541716351f * th + 0.7139816917074209f);
    b[1] = 0.1017f * t2 * (t2 - 2.1966568338249754f * th + 1.2949852507374631f);
    b[2] = 2.490627285651252793f * t2 * (t2 - 2.38535645472061657f * th + 1.57803468208092486f);
    b[3] = -16.54810288924490272f * (th - 1.21712927295533244f) * (th - 0.61620406037800089f) * t2;
    b[4] = 47.37952196281928122f * (th - 1.203071208372362603f) * (th - 0.658047292653547382f) * t2;
    b[5] = -34.87065786149660974f * (th - 1.2f) * (th - 0.666666666666666667f) * t2;
    b[6] = 2.5f * (th - 1.0f) * (th - 0.6f) * t2;
}
// d b_i / d theta (analytic), for the theta-cotangent of saveat points in the reverse pass
__host__ __device__ inline void dense_weights_deriv(float th, float (&db)[7]) {
    const float t2 = th * th;
    {   // b1 = c * f * g, f = th^2 - r th, g = th^2 - p th + q
        const float c = -1.0530884977290216f, r = 1.3299890189751412f, p = 1.4364028541716351f, q = 0.7139816917074209f;
        const float f = t2 - r * th, g = t2 - p * th + q;
        db[0] = c * ((2.f * th - r) * g + f * (2.f * th - p));
    }
    {   // b = c * th^2 * (th^2 - p th + q)
        const float c[2] = {0.1017f, 2.490627285651252793f};
        const float p[2] = {2.1966568338249754f, 2.38535645472061657f}, q[2] = {1.2949852507374631f, 1.57803468208092486f};
        for (int i = 0; i < 2; ++i) db[1 + i] = c[i] * (2.f * th * (t2 - p[i] * th + q[i]) + t2 * (2.f * th - p[i]));
    }
    {   // b = c * (th - r)(th - s) * th^2
        const float c[4] = {-16.54810288924490272f, 47.37952196281928122f, -34.87065786149660974f, 2.5f};
        const float r[4] = {1.21712927295533244f, 1.203071208372362603f, 1.2f, 1.0f};
        const float s[4] = {0.61620406037800089f, 0.658047292653547382f, 0.666666666666666667f, 0.6f};
        for (int i = 0; i < 4; ++i) {
            const float f = (th - r[i]) * (th - s[i]);
            db[3 + i] = c[i] * ((2.f * th - r[i] - s[i]) * t2 + f * 2.f * th);
        }
    }
}
// PI controller constants
constexpr float kBeta1 = (float)(7.0 / 50.0);
constexpr float kBeta2 = (float)(2.0 / 25.0);
constexpr float kGamma = 0.9f;
constexpr float kQmin = 0.2f;
constexpr float kQmax = 10.0f;
constexpr float kQoldInit = 1e-4f;
constexpr float kDtMin = 1.1920929e-7f;  // eps(Float32)

template <int NG>
struct Geo {
    static constexpr int BT = 4 * NG;        // batch columns per workgroup
    static constexpr int RG = 16 / NG;       // row groups per MFMA
    static constexpr int TR = 64 / NG;       // rows per MFMA tile
    static constexpr int MTS = 128 / TR;     // max tiles of the "small-M" GEMM (M <= 128)
    static constexpr int TPW = (NG == 1) ? 2 : 4;  // tiles per wave of the "big-M" GEMM (M <= 8*TPW*TR)
};

// flags in StepMeta
enum : int { F_ACCEPT = 1, F_CLAMP = 2, F_QCLAMP = 4, F_EZERO = 8, F_DTMAXCLAMP = 16, F_REJQ11 = 32 };

struct StepState {  // state BEFORE an attempt (double-buffered in HBM, one writer: workgroup 0)
    float t, dtp, qold, last_eest;
    int live;    // tape record holding the current (uprev, k1); -1 = (x, f0)
    int done;    // integration finished or aborted
    int status;  // rnde_status of the solve
    int n_att, n_acc;
    int next_save;   // saveat: index of the first save time not yet written
    int pad[2];
};
struct StepMeta {  // one per attempted step; consumed by the reverse pass and by the host
    float t, dt, dtp_in, eest, q11, q, qold_in, rej_m;
    int flags, src, rec;
    float eigen, n1, n2;   // stiffness estimate ||k7-k6|| / ||unew-g6|| and its two norms (regularize >= 2 only)
    int pad[2];
};
struct InitRec {  // initial-step heuristic record
    float d0, d1, d2, dt0, dt1, dt;
    int dt0_const, dt0_clamped, sel, dt1_const, max_is_d2, pad;
};

struct StepParams {
    const float* x;  // D x B, caller layout
    float* f0; float* h0; float* u1; float* f1; float* h1;  // D x Bpad / H x Bpad
    float* arena; long long rec_stride;                      // tape records
    StepState* ctl;     // [2]
    StepState* ctl_final;
    StepMeta* meta;
    InitRec* initrec;
    float* errpart;     // [2][nwg]
    float* initpart;    // [3][nwg]
    float* dbg_out;     // FEVAL output (caller layout) / finish copy target
    int D, H, B, Bpad, nwg;
    int Bn;             // columns the norms are means over: B, or the GLOBAL batch when the controller is shared by several shards (SURVEY 8e mode 2)
    float reltol, abstol, t0, t1;
    int tape, max_attempts;
    int forced; float forced_t, forced_dt;  // debug/bench: run one attempt from a given (t, dt)
    int rec_shift;                           // bench (forced attempts, n = 0): tape record the attempt writes -- cycled, so that the tape is as cold as in a solve
    int xvec;                                // x is 16-byte aligned and D % 4 == 0
    int reg_kind;                            // rnde_reg: 2, 3 also need the stiffness estimate
    const float* sv_t; int nsave; float* sv_out;   // saveat times (device), their count, output D x T x B
    // replay (rnde_node_forward_replay): attempt n takes the proposed size replay[2n] and the accept decision replay[2n+1] != 0
    // instead of the controller's; the solve ends after n_replay attempts.  EEst, q11, q are still computed and recorded.
    const float* replay; int n_replay;
    // step-size controller exponents and the order in the initial-step rule: 7 / (10 order), 2 / (5 order), order -- kBeta1, kBeta2, 5 for
    // the order-5 pairs (every reference call site); another order only with a pair that comes as a table (RkTab.order, rnde_chainmw.h)
    float beta1, beta2, rk_order;
    // large batches (round 6, stage engine's two-tile attempt kernel): the three cross-workgroup sums over errpart[parity], formed ONCE behind the launch that wrote
    // them (rnde_epart_reduce_kernel: sum_partials, the same additions in the same order) -- [2][4] doubles, or NULL: every workgroup's prologue sums for itself
    const double* esum;
};

// record layout inside the arena (floats): k2..k7 | g2..g6 | unew | h2..h7 | z1bar2..z1bar7
// (the reverse pass overwrites k_s in place with the layer-2 pre-activation cotangent z2bar_s once
//  k_s is dead, and stores the layer-1 one in z1)
struct RecLayout {
    long long A, HB;
    __host__ __device__ long long k(int s) const { return (long long)(s - 2) * A; }       // s = 2..7
    __host__ __device__ long long g(int s) const { return (long long)(6 + s - 2) * A; }   // s = 2..6
    __host__ __device__ long long unew() const { return 11LL * A; }
    __host__ __device__ long long h(int s) const { return 12LL * A + (long long)(s - 2) * HB; }
    __host__ __device__ long long z1(int s) const { return 12LL * A + 6LL * HB + (long long)(s - 2) * HB; }
    // stage engine: the attempt's own copies of (uprev, k1), so that later launches of the attempt can issue their
    // loads from host-known addresses without waiting for the controller state (which record is "live")
    __host__ __device__ long long upc() const { return 12LL * A + 12LL * HB; }
    __host__ __device__ long long k1c() const { return 13LL * A + 12LL * HB; }
    __host__ __device__ long long total() const { return 14LL * A + 12LL * HB; }
};

// Sum over the 64 lanes of a wave, the same value returned to every lane.  DPP moves (row_shr 1, 2, 4, 8 inside each row of 16
// lanes, then row_bcast 15 and 31 across the rows) leave the total in lane 63, read back through a scalar register: six short VALU
// steps.  (The __shfl_xor butterfly these replace is six dependent ds_bpermute round trips through the LDS pipe, ~700 cycles per sum;
// the controller of every launch does three to four of them in double precision before anything else can start.)  Fixed order of
// additions: the same bits on every lane, workgroup and launch.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get_f(float v) {      // lanes without a source (or outside ROW_MASK) read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double s) {
    s += dpp_get_d<0x111, 0xf>(s);      // row_shr:1
    s += dpp_get_d<0x112, 0xf>(s);      // row_shr:2
    s += dpp_get_d<0x114, 0xf>(s);      // row_shr:4
    s += dpp_get_d<0x118, 0xf>(s);      // row_shr:8   -> lane 15 of each row holds the row's sum
    s += dpp_get_d<0x142, 0xa>(s);      // row_bcast:15 into rows 1 and 3
    s += dpp_get_d<0x143, 0xc>(s);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float wave_sum_f(float s) {
    s += dpp_get_f<0x111, 0xf>(s);
    s += dpp_get_f<0x112, 0xf>(s);
    s += dpp_get_f<0x114, 0xf>(s);
    s += dpp_get_f<0x118, 0xf>(s);
    s += dpp_get_f<0x142, 0xa>(s);
    s += dpp_get_f<0x143, 0xc>(s);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
}
// fixed-order sum of n fp32 partials, carried in double; identical result on every lane / workgroup
// (four loads per lane are requested before the first add: the plain loop `for (i = lane; i < n; i += 64) s += part[i]` waits
//  for each cold load in turn -- 8 k cycles in front of every launch for n = 224; same additions in the same order)
// pre: the first block's four values of this lane, requested by the caller earlier (partials_request), or nullptr
__device__ __forceinline__ void partials_request(const float* __restrict__ part, int lane, float (&v)[4]) {
    v[0] = part[lane]; v[1] = part[lane + 64]; v[2] = part[lane + 128]; v[3] = part[lane + 192];
}
__device__ __forceinline__ double sum_partials(const float* __restrict__ part, int n, int lane, const float (*pre)[4] = nullptr) {
    double s = 0;
    for (int base = 0; base < n; base += 256) {
        const int i0 = base + lane, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
        // unconditional requests of the whole 256-entry block (the arrays are padded by 256 entries; values past n are not added): loads
        // under a lane condition were compiled into two dependent cold round trips, clamped indices into four address computations
        float v0, v1, v2, v3;
        if (base == 0 && pre) {   // (the empty asm is where the values are first USED: without it the compiler converts them to double right behind
            v0 = (*pre)[0]; v1 = (*pre)[1]; v2 = (*pre)[2]; v3 = (*pre)[3];   //  the loads and waits for them there, in front of everything issued after)
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
        else { v0 = part[i0]; v1 = part[i1]; v2 = part[i2]; v3 = part[i3]; }
        if (i0 < n) s += (double)v0;
        if (i1 < n) s += (double)v1;
        if (i2 < n) s += (double)v2;
        if (i3 < n) s += (double)v3;
    }
    return wave_sum_d(s);
}

// ---- weight streaming: per-wave LDS-DMA ring ------------------------------------------------------
// The Dense-layer GEMMs of one workgroup re-stream all packed weights (~0.7 MB) from L2 for every f
// evaluation, and each weight is used for only BT columns, so the loop is bound by bytes in flight, not
// by MFMA issue: with register prefetch a wave can keep ~2 KB in flight (measured 27 GB/s per CU).
// Instead every wave owns kRing 1-KiB slots of LDS and keeps kRing `global_load_lds_dwordx4`
// (1 KiB each, no VGPR destination) in flight: 8 waves x 12 KiB = 96 KiB per CU.  The ring is private to
// the wave, so no workgroup barrier is involved: a counted s_waitcnt vmcnt orders DMA -> ds_read, and
// s_waitcnt lgkmcnt(0) orders ds_read -> slot reuse.
// One ring unit = 64 lanes x 16 B = (one tile, KU = 64/TR consecutive k4 groups) of the packed weights.
constexpr int kRing = 12;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ void dma_unit(const f32x4* __restrict__ gsrc_lane, float* lds_slot_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc_lane, (lds_void_t*)lds_slot_uniform, 16, 0, 0);
}
// the same with the non-temporal hint (a stream that is read once: keep it from displacing what a neighbouring kernel keeps in L2)
__device__ __forceinline__ void dma_unit_nt(const f32x4* __restrict__ gsrc_lane, float* lds_slot_uniform) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc_lane, (lds_void_t*)lds_slot_uniform, 16, 0, 2);   // aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float act_apply(int act, float v) { return act ? tanhf(v) : v; }

// tanh for the stage engine: ~15 VALU instead of ocml's ~45 (the element-wise phases of those kernels are
// instruction-issue bound).  |x| < 0.55: odd polynomial x + x^3 P(x^2) (least-squares fit, 5 terms);
// otherwise 1 - 2/(exp(2|x|)+1) with a compensated exp2 argument and one Newton step on the reciprocal.
// Max error 1.65 ulp (mean 0.27) against fp64 on 5e6 samples with exact exp2/division (libm tanhf: 1.37);
// the hardware v_exp_f32 / v_rcp_f32 add at most ~1 ulp.  Accuracy matters beyond parity: at the reference's
// tolerance the step size is set by rounding noise, so a sloppy tanh would RAISE NFE.
__device__ __forceinline__ float tanh_fast(float x) {
    // (|x| clamped at 9.1 instead of a select behind the formula: the formula's value AT 9.1 is exactly 1.0f, so every result is the same
    //  bit pattern as with the select, and |x| + clamp is ONE v_min_f32 with a source modifier where abs, compare and select were three)
    const float ax = fminf(fabsf(x), 9.1f), x2 = x * x;
    float p = -0.00671552f;
    p = fmaf(p, x2, 0.02136713f);
    p = fmaf(p, x2, -0.05391917f);
    p = fmaf(p, x2, 0.13333165f);
    p = fmaf(p, x2, -0.33333332f);
    const float small = fmaf(x, x2 * p, x);
    constexpr float L = 2.8853900817779268f;                                   // 2 log2(e)
    constexpr float Llo = (float)(2.8853900817779268 - (double)L);
    const float yh = ax * L;
    const float yl = fmaf(ax, L, -yh) + ax * Llo;
    float e = __builtin_amdgcn_exp2f(yh);
    e = fmaf(e, yl * 0.6931471805599453f, e);
    const float dd = e + 1.0f;
    float r = __builtin_amdgcn_rcpf(dd);
    r = fmaf(fmaf(-dd, r, 1.0f), r, r);
    const float big = fmaf(-2.0f, r, 1.0f);
    return ax < 0.55f ? small : copysignf(big, x);
}
// two values per instruction (v_pk_fma_f32 / v_pk_mul_f32): same operation sequence per component as tanh_fast, so
// results are bit-identical to it
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 tanh_fast2(f32x2 x) {
    const f32x2 ax = {fminf(fabsf(x.x), 9.1f), fminf(fabsf(x.y), 9.1f)}, x2 = x * x;
    f32x2 p = fma2((f32x2)(-0.00671552f), x2, (f32x2)(0.02136713f));
    p = fma2(p, x2, (f32x2)(-0.05391917f));
    p = fma2(p, x2, (f32x2)(0.13333165f));
    p = fma2(p, x2, (f32x2)(-0.33333332f));
    const f32x2 sm = fma2(x, x2 * p, x);
    constexpr float L = 2.8853900817779268f;
    constexpr float Llo = (float)(2.8853900817779268 - (double)L);
    const f32x2 yh = ax * L;
    const f32x2 yl = fma2(ax, (f32x2)(L), -yh) + ax * Llo;
    f32x2 e = {__builtin_amdgcn_exp2f(yh.x), __builtin_amdgcn_exp2f(yh.y)};
    e = fma2(e, yl * 0.6931471805599453f, e);
    const f32x2 dd = e + 1.0f;
    f32x2 r = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
    r = fma2(fma2(-dd, r, (f32x2)(1.0f)), r, r);
    const f32x2 big = fma2((f32x2)(-2.0f), r, (f32x2)(1.0f));
    f32x2 o;
    o.x = ax.x < 0.55f ? sm.x : copysignf(big.x, x.x);
    o.y = ax.y < 0.55f ? sm.y : copysignf(big.y, x.y);
    return o;
}
__device__ __forceinline__ float act_apply_fast(int act, float v) { return act ? tanh_fast(v) : v; }

}  // namespace rnde
