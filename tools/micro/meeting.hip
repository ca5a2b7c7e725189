// micro-benchmark: the cross-XCD MEETING of the one-launch solve (csrc/rnde_stage_solve.h `solve_meet`): 224 workgroups, one per CU, each contributes one
// float per round; every workgroup needs the sum of all of them (in a FIXED order) before it goes on.  The product's form (FLAT) measured 2.2-3.0 us from the
// LAST arrival to the way out (profiles/r05_attempt_stamps.txt) -- slower than a kernel boundary.  Which form is faster?
//   FLAT        every workgroup publishes an 8-byte {value, tag} granule (agent-scope store) and polls ALL granules (agent-scope loads): 224 pollers x 14 lines
//   XPOLL       publish as FLAT; only the 8 XCD leaders poll the 224 granules and form the sums; a leader hands the result to the workgroups of its XCD through
//               the XCD's L2 (plain 16-byte {double, tag} store, sc1-load polls) -- same additions in the same order as FLAT: bit-identical sums
//   TREE        members publish into their XCD's L2 (plain store); the leader polls its 28, adds them, publishes ONE agent-scope granule per XCD; every
//               workgroup polls the 8 XCD granules (one line) and adds them in XCD order (a DIFFERENT association than FLAT)
//   TREE_L      as TREE, but only the leaders poll the 8 XCD granules and broadcast through L2 (three hops, least memory-side traffic)
// Each round r uses fresh granules (index r), as the product does.  Arrival skew: workgroup b waits (hash(b, r) % skew) cycles before it arrives.
// Output per variant: mean round time without skew, and with skew the time from the LAST arrival to the first / median / last way out (100 MHz clock).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
enum { FLAT = 0, XPOLL = 1, TREE = 2, TREE_L = 3 };
constexpr int NWG = 224, ROUNDS_REC = 64;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
struct Bufs {
    unsigned long long* flat;      // [rounds][256] granules {value, tag}
    u32x4* bcast;                  // [rounds][8][8] 16-byte {double, tag, tag} per XCD (a 128-byte line of its own per XCD and round)
    u32x4* local;                  // [rounds][8][32] XCD-local 16-byte entries {value, tag, tag, tag} (TREE): one store, one b128 load -- never torn
    unsigned long long* xsum;      // [rounds][8 (+pad to 16)] per-XCD sums, agent scope (TREE)
    unsigned* abort_flag;
    unsigned long long* stamps;    // [ROUNDS_REC][NWG][2] arrive / out
    double* result;                // [NWG] last sum seen
    unsigned* xcc;                 // [NWG]
};

__device__ __forceinline__ bool poll_flat(const unsigned long long* base, unsigned tag, int lane, unsigned* abort_flag, double& out) {
    unsigned long long e[4] = {0, 0, 0, 0};
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ok[q] = lane + 64 * q >= NWG;
    int spins = 0;
    while (true) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (!ok[q]) { e[q] = __hip_atomic_load(base + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok[q] = (unsigned)(e[q] >> 32) == tag; }
        if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
        if (++spins > 2000000) { if (lane == 0) atomicExch(abort_flag, 1u); return false; }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < NWG) s += (double)__uint_as_float((unsigned)(e[q] & 0xFFFFFFFFull));
    out = wave_sum_d(s);
    return true;
}
// one 16-byte record {double, tag, tag} through the XCD's L2: plain store by the leader, sc1-load polls by the members (the slab hand-off's protocol)
__device__ __forceinline__ void bcast_put(u32x4* rec, double v, unsigned tag, int lane) {
    if (lane == 0) { const unsigned long long b = __double_as_longlong(v); *rec = (u32x4){(unsigned)b, (unsigned)(b >> 32), tag, tag}; }
}
__device__ __forceinline__ bool bcast_poll(const u32x4* rec, unsigned tag, unsigned* abort_flag, double& out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, 16, 0x00020000);
    int spins = 0;
    while (true) {
        __asm__ volatile("" ::: "memory");
        const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 16);      // sc1: misses L1, served by the XCD's L2
        if (e[2] == tag && e[3] == tag) { out = __longlong_as_double(((unsigned long long)e[1] << 32) | e[0]); return true; }
        if (++spins > 2000000) { atomicExch(abort_flag, 1u); return false; }
    }
}

template <int MODE>
__global__ __launch_bounds__(448) void meeting_kernel(Bufs Z, int rounds, int skew, unsigned epoch) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int xg = wg & 7;                       // workgroups with the same blockIdx % 8 share an XCD (round-robin dispatch; checked on the host from xcc[])
    const int member = wg >> 3;                  // 0..27 within the XCD group
    const bool leader = member == 0;
    __shared__ double sh[2];
    if (tid == 0) Z.xcc[wg] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
    double res = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = epoch * 65536u + (unsigned)r + 1u;
        if (skew > 0) {      // arrival skew: a per-(workgroup, round) delay
            const unsigned hsh = (unsigned)(wg * 2654435761u) ^ (unsigned)(r * 40503u);
            const unsigned long long until = clock64() + (hsh % (unsigned)skew);
            while (clock64() < until) { }
        }
        __syncthreads();
        const float mine = 1.0f + 1e-3f * (float)((wg * 7 + r) & 255);
        if (w == 0) {
            if (lane == 0 && r < ROUNDS_REC) Z.stamps[((size_t)r * NWG + wg) * 2] = wall_clock64();
            bool ok = true;
            double s = 0;
            if (MODE == FLAT || MODE == XPOLL) {
                unsigned long long* base = Z.flat + (size_t)r * 256;
                if (lane == 0) __hip_atomic_store(base + wg, ((unsigned long long)tag << 32) | __float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (MODE == FLAT) ok = poll_flat(base, tag, lane, Z.abort_flag, s);
                else {
                    u32x4* rec = Z.bcast + ((size_t)r * 8 + xg) * 8;
                    if (leader) { ok = poll_flat(base, tag, lane, Z.abort_flag, s); bcast_put(rec, s, tag, lane); }
                    else ok = bcast_poll(rec, tag, Z.abort_flag, s);
                }
            } else {      // TREE / TREE_L
                u32x4* loc = Z.local + ((size_t)r * 8 + xg) * 32;
                unsigned long long* xs = Z.xsum + (size_t)r * 16;
                if (lane == 0) loc[member] = (u32x4){__float_as_uint(mine), tag, tag, tag};      // plain store: stays in this XCD's L2
                if (leader) {
                    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)loc, 0, 512, 0x00020000);
                    int spins = 0;
                    unsigned e = 0;
                    while (true) {
                        __asm__ volatile("" ::: "memory");
                        u32x4 v = {0u, tag, tag, tag};
                        if (lane < 28) v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 16);
                        e = v[0];
                        if (__all(v[1] == tag && v[3] == tag)) break;
                        if (++spins > 2000000) { if (lane == 0) atomicExch(Z.abort_flag, 1u); ok = false; break; }
                    }
                    const float xsum = (float)wave_sum_d(lane < 28 ? (double)__uint_as_float(e) : 0.0);
                    if (lane == 0) __hip_atomic_store(xs + xg, ((unsigned long long)tag << 32) | __float_as_uint(xsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (MODE == TREE || leader) {
                    int spins = 0;
                    unsigned long long e = 0;
                    bool got = lane >= 8;
                    while (ok) {
                        if (!got) { e = __hip_atomic_load(xs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got = (unsigned)(e >> 32) == tag; }
                        if (__all(got)) break;
                        if (++spins > 2000000) { if (lane == 0) atomicExch(Z.abort_flag, 1u); ok = false; }
                    }
                    s = wave_sum_d(lane < 8 ? (double)__uint_as_float((unsigned)e) : 0.0);
                    if (MODE == TREE_L) bcast_put(Z.bcast + ((size_t)r * 8 + xg) * 8, s, tag, lane);
                } else ok = bcast_poll(Z.bcast + ((size_t)r * 8 + xg) * 8, tag, Z.abort_flag, s);
            }
            if (lane == 0 && r < ROUNDS_REC) Z.stamps[((size_t)r * NWG + wg) * 2 + 1] = wall_clock64();
            if (lane == 0) { sh[0] = s; sh[1] = ok ? 0.0 : 1.0; }
        }
        __syncthreads();
        res = sh[0];
        if (sh[1] != 0.0) break;
    }
    if (tid == 0) Z.result[wg] = res;
}

template <int MODE>
static void run(const char* name, Bufs Z, int rounds, FILE* csv) {
    static unsigned epoch = 1;
    for (int skew : {0, 3000}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(meeting_kernel<MODE>, dim3(NWG), dim3(448), 0, 0, Z, rounds, skew, epoch++);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(meeting_kernel<MODE>, dim3(NWG), dim3(448), 0, 0, Z, rounds, skew, epoch++);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st((size_t)ROUNDS_REC * NWG * 2);
        std::vector<double> res(NWG);
        std::vector<unsigned> xcc(NWG);
        unsigned ab = 0;
        CK(hipMemcpy(st.data(), Z.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(res.data(), Z.result, NWG * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xcc.data(), Z.xcc, NWG * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ab, Z.abort_flag, 4, hipMemcpyDeviceToHost));
        bool placed = true;
        for (int b = 8; b < NWG; ++b) placed = placed && xcc[b] == xcc[b & 7];
        // the expected sum of the last round
        double want = 0;
        for (int b = 0; b < NWG; ++b) want += (double)(1.0f + 1e-3f * (float)((b * 7 + rounds - 1) & 255));
        bool same = true;
        for (int b = 1; b < NWG; ++b) same = same && res[b] == res[0];
        std::vector<double> first, med, last, spread;
        for (int r = 8; r < std::min(rounds, ROUNDS_REC); ++r) {
            unsigned long long la = 0, fa = ~0ull;
            std::vector<unsigned long long> outs;
            for (int b = 0; b < NWG; ++b) { la = std::max(la, st[((size_t)r * NWG + b) * 2]); fa = std::min(fa, st[((size_t)r * NWG + b) * 2]); outs.push_back(st[((size_t)r * NWG + b) * 2 + 1]); }
            std::sort(outs.begin(), outs.end());
            first.push_back(((double)outs[0] - (double)la) * 0.01); med.push_back(((double)outs[NWG / 2] - (double)la) * 0.01); last.push_back(((double)outs[NWG - 1] - (double)la) * 0.01);
            spread.push_back((double)(la - fa) * 0.01);
        }
        auto mean = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return v.empty() ? 0.0 : s / v.size(); };
        printf("%-8s skew %4d cycles: %6.2f us per round | arrival spread %5.2f us | out after the LAST arrival: first %5.2f median %5.2f last %5.2f us | abort %u placement %s sums %s (%.6f vs %.6f)\n",
               name, skew, 1e3 * ms / rounds, mean(spread), mean(first), mean(med), mean(last), ab, placed ? "ok" : "NOT by blockIdx%8", same ? "equal" : "DIFFER", res[0], want);
        if (csv) fprintf(csv, "%s,%d,%.3f,%.3f,%.3f,%.3f,%.3f,%u,%d,%d\n", name, skew, 1e3 * ms / rounds, mean(spread), mean(first), mean(med), mean(last), ab, (int)placed, (int)same);
    }
}

int main(int argc, char** argv) {
    const int rounds = 400;
    Bufs Z;
    CK(hipMalloc(&Z.flat, (size_t)rounds * 256 * 8)); CK(hipMalloc(&Z.bcast, (size_t)rounds * 8 * 8 * 16)); CK(hipMalloc(&Z.local, (size_t)rounds * 8 * 32 * 16));
    CK(hipMalloc(&Z.xsum, (size_t)rounds * 16 * 8)); CK(hipMalloc(&Z.abort_flag, 4)); CK(hipMalloc(&Z.stamps, (size_t)ROUNDS_REC * NWG * 16));
    CK(hipMalloc(&Z.result, NWG * 8)); CK(hipMalloc(&Z.xcc, NWG * 4));
    CK(hipMemset(Z.flat, 0, (size_t)rounds * 256 * 8)); CK(hipMemset(Z.bcast, 0, (size_t)rounds * 8 * 8 * 16)); CK(hipMemset(Z.local, 0, (size_t)rounds * 8 * 32 * 16));
    CK(hipMemset(Z.xsum, 0, (size_t)rounds * 16 * 8)); CK(hipMemset(Z.abort_flag, 0, 4));
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "variant,skew_cycles,us_per_round,arrival_spread_us,out_first_us,out_median_us,out_last_us,abort,placement_ok,sums_equal\n");
    for (int rep = 0; rep < 2; ++rep) {
        run<FLAT>("FLAT", Z, rounds, csv);
        run<XPOLL>("XPOLL", Z, rounds, csv);
        run<TREE>("TREE", Z, rounds, csv);
        run<TREE_L>("TREE_L", Z, rounds, csv);
    }
    if (csv) fclose(csv);
    return 0;
}
