// micro-benchmark: hand-off of a split-K slab between the R = 7 workgroups of a column tile (same blockIdx % 8, i.e.
// same XCD under round-robin dispatch) INSIDE one kernel: stores -> workgroup-release -> relaxed agent-scope flag;
// consumer: poll flags (L1 bypass), agent-scope acquire (buffer_inv sc1), plain loads.  All spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int R = 7, C = 32, HT = 7, WT = 7;
__global__ __launch_bounds__(448) void k(f32x4* slab, unsigned* flags, unsigned* abort_flag, unsigned* xcc, unsigned long long* st, float* out, int iters, unsigned base) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = blockIdx.x / C, ct = blockIdx.x - rb * C;
    if (tid == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
    f32x4 acc = {1.f, 2.f, 3.f, 4.f};
    unsigned long long c0 = clock64();
    bool dead = false;
    for (int it = 0; it < iters && !dead; ++it) {
        const int par = it & 1;
        // produce: this block's partial for every hidden tile (one f32x4 per lane per wave)
        f32x4* sl = slab + ((((size_t)par * C + ct) * R + rb) * HT) * 64;
        sl[(size_t)w * 64 + lane] = acc * (1.f + 1e-3f * rb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + ct * 8 + rb, base + it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // consume
        if (w == 0) {
            unsigned v = base + it + 1;
            int spins = 0;
            while (true) {
                if (lane < R) v = __hip_atomic_load(flags + ct * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (lane >= R) || ((int)(v - (base + it + 1)) >= 0);
                if (__all(ok)) break;
                if (++spins > 200000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { dead = true; break; }
        const f32x4* s0 = slab + (((size_t)par * C + ct) * R) * HT * 64;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) z += s0[((size_t)r * HT + w) * 64 + lane];
        acc = z * 0.1428f;
    }
    unsigned long long c1 = clock64();
    out[blockIdx.x * 448 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) st[blockIdx.x] = c1 - c0;
}
int main() {
    f32x4* slab; unsigned *flags, *abortf, *xcc; unsigned long long* st; float* out;
    const int G = R * C;
    hipMalloc(&slab, (size_t)2 * C * R * HT * 64 * 16); hipMalloc(&flags, C * 8 * 4); hipMalloc(&abortf, 4); hipMalloc(&xcc, G * 4);
    hipMalloc(&st, G * 8); hipMalloc(&out, (size_t)G * 448 * 4);
    hipMemset(flags, 0, C * 8 * 4); hipMemset(abortf, 0, 4); hipMemset(slab, 0, (size_t)2 * C * R * HT * 64 * 16);
    const int iters = 2000;
    unsigned base = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(G), dim3(448), 0, 0, slab, flags, abortf, xcc, st, out, iters, base);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        base += iters;
        unsigned ab; hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> hx(G); hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
        std::vector<float> ho((size_t)G * 448); hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int ct = 0; ct < C; ++ct) for (int rb = 1; rb < R; ++rb) if (hx[rb * C + ct] != hx[ct]) ++bad;
        printf("rep %d: %.3f us per hand-off (kernel %.2f ms), abort=%u, clusters with mixed XCC: %d, xcc of wg 0..9:", rep, ms * 1e3 / iters, ms, ab, bad);
        for (int i = 0; i < 10; ++i) printf(" %u", hx[i]);
        printf("  out[0]=%g out[last]=%g\n", ho[0], ho.back());
    }
    return 0;
}
