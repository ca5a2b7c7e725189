// How much of a hand-off is the polling volume?  The persistent kernels' tagged entries (two 16-byte entries per lane and tile:
// {v0, v1, tag, tag}, {v2, v3, tag, tag}) against a stand-in with ONE 16-byte entry per lane and tile ({v0, v1, v2, tag}: what a
// tag-free sentinel protocol would move).  Same grid, same barrier per exchange.  Prints microseconds per exchange.
#include "../../regneuralde.jl_amd/csrc/rnde_stage_persist.h"
#include <cstdio>
using namespace rnde;
constexpr int R = 7, C = 32, HT = 7;
struct Sync { float* tslab; unsigned* abort_flag; unsigned* xcc; unsigned seq_base; int max_spins; };
#define PersistSync Sync
// the previous protocol of the product kernels: two tagged entries per lane and tile
__device__ __forceinline__ void slab_put_tagged(float* tslab, size_t tile_index, int lane, const f32x4& v, unsigned tag) {
    const float tf = __builtin_bit_cast(float, tag);
    f32x4* d = (f32x4*)tslab + tile_index * 128;
    d[lane] = (f32x4){v[0], v[1], tf, tf};
    d[64 + lane] = (f32x4){v[2], v[3], tf, tf};
}
__device__ __forceinline__ bool slab_poll_sum_tagged(const Sync& Y, int par, int ct, int ht, int lane, unsigned tag, f32x4& zs) {
    const float* base = Y.tslab + ((((size_t)par * C + ct) * R) * HT) * 512;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    while (true) {
        __asm__ volatile("" ::: "memory");
        u32x4 e0[R], e1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int off = ((r * HT + ht) * 128 + lane) * 16;
            e0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
            e1[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024, 0, 16);
        }
        bool ok = true;
#pragma unroll
        for (int r = 0; r < R; ++r) ok = ok && e0[r][2] == tag && e0[r][3] == tag && e1[r][2] == tag && e1[r][3] == tag;
        if (__all(ok)) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) { const f32x4 f0 = __builtin_bit_cast(f32x4, e0[r]), f1 = __builtin_bit_cast(f32x4, e1[r]); s0 += f0[0]; s1 += f0[1]; s2 += f1[0]; s3 += f1[1]; }
            zs = (f32x4){s0, s1, s2, s3};
            return true;
        }
        if (++spins > Y.max_spins) return false;
    }
}
// indicator first: spin on ONE dword per producer (word 3 of lane 63's entry), then one full load (re-validated)
__device__ __forceinline__ bool poll_ind(const Sync& Y, int par, int ct, int ht, int lane, unsigned tag, f32x4& zs) {
    const float* base = Y.tslab + ((((size_t)par * C + ct) * R) * HT) * 512;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    while (true) {
        __asm__ volatile("" ::: "memory");
        const int r = lane < R ? lane : 0;
        const unsigned ind = __builtin_amdgcn_raw_buffer_load_b32(rs, ((r * HT + ht) * 128 + 63) * 16 + 12, 0, 16);
        if (__all(ind == tag)) break;
        if (++spins > Y.max_spins) return false;
    }
    while (true) {
        __asm__ volatile("" ::: "memory");
        u32x4 e0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) e0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r * HT + ht) * 128 + lane) * 16, 0, 16);
        bool ok = true;
#pragma unroll
        for (int r = 0; r < R; ++r) ok = ok && e0[r][3] == tag;
        if (__all(ok)) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) { const f32x4 f = __builtin_bit_cast(f32x4, e0[r]); s0 += f[0]; s1 += f[1]; s2 += f[2]; }
            zs = (f32x4){s0, s1, s2, 0.f};
            return true;
        }
        if (++spins > Y.max_spins) return false;
    }
}
__device__ __forceinline__ bool poll1(const PersistSync& Y, int par, int ct, int ht, int lane, unsigned tag, f32x4& zs) {
    const float* base = Y.tslab + ((((size_t)par * C + ct) * R) * HT) * 512;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    while (true) {
        __asm__ volatile("" ::: "memory");   // the polling loads must stay in the loop
        u32x4 e0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) e0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r * HT + ht) * 128 + lane) * 16, 0, (int)0x80000010);
        bool ok = true;
#pragma unroll
        for (int r = 0; r < R; ++r) ok = ok && e0[r][3] == tag;
        if (__all(ok)) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) { const f32x4 f = __builtin_bit_cast(f32x4, e0[r]); s0 += f[0]; s1 += f[1]; s2 += f[2]; }
            zs = (f32x4){s0, s1, s2, 0.f};
            return true;
        }
        if (++spins > Y.max_spins) return false;
    }
}
__device__ __forceinline__ bool poll1c(const PersistSync& Y, int par, int ct, int ht, int lane, unsigned tag, f32x4& zs) {
    const float* base = Y.tslab + ((((size_t)par * C + ct) * R) * HT) * 256;     // compact: 64 f32x4 = 256 floats per tile
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int spins = 0;
    while (true) {
        __asm__ volatile("" ::: "memory");   // the polling loads must stay in the loop
        u32x4 e0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) e0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r * HT + ht) * 64 + lane) * 16, 0, (int)0x80000010);
        bool ok = true;
#pragma unroll
        for (int r = 0; r < R; ++r) ok = ok && e0[r][3] == tag;
        if (__all(ok)) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) { const f32x4 f = __builtin_bit_cast(f32x4, e0[r]); s0 += f[0]; s1 += f[1]; s2 += f[2]; }
            zs = (f32x4){s0, s1, s2, 0.f};
            return true;
        }
        if (++spins > Y.max_spins) return false;
    }
}
template <int ONE>
__global__ __launch_bounds__(448) void k(PersistSync Y, float* sink, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = (blockIdx.x >> 3) % R, ct = 8 * ((blockIdx.x >> 3) / R) + (blockIdx.x & 7);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        const unsigned tag = Y.seq_base + it + 1;
        const size_t tile0 = (((size_t)par * C + ct) * R + rb) * HT;
        const f32x4 v = {(float)it, acc[1] * 1e-9f + 1.f, 2.f, 3.f};
        f32x4 zs;
        if (ONE == 1) {
            ((f32x4*)Y.tslab + (tile0 + w) * 128)[lane] = (f32x4){v[0], v[1], v[2], __builtin_bit_cast(float, tag)};
            if (!poll1(Y, par, ct, w, lane, tag, zs)) break;
        } else if (ONE == 4) {       // one entry, tiles packed 1 KB apart
            ((f32x4*)Y.tslab + (tile0 + w) * 64)[lane] = (f32x4){v[0], v[1], v[2], __builtin_bit_cast(float, tag)};
            if (!poll1c(Y, par, ct, w, lane, tag, zs)) break;
        } else if (ONE == 5) {       // one entry, indicator-first polling
            ((f32x4*)Y.tslab + (tile0 + w) * 128)[lane] = (f32x4){v[0], v[1], v[2], __builtin_bit_cast(float, tag)};
            if (!poll_ind(Y, par, ct, w, lane, tag, zs)) break;
        } else if (ONE == 2) {       // two entries written, one polled
            const float tf = __builtin_bit_cast(float, tag);
            ((f32x4*)Y.tslab + (tile0 + w) * 128)[lane] = (f32x4){v[0], v[1], v[2], tf};
            ((f32x4*)Y.tslab + (tile0 + w) * 128)[64 + lane] = (f32x4){v[0], v[1], v[2], tf};
            if (!poll1(Y, par, ct, w, lane, tag, zs)) break;
        } else if (ONE == 3) {       // one entry written ({v0, v1, tag, tag} form), the tagged poll on a slab whose second entries carry the tag from a pre-fill
            const float tf = __builtin_bit_cast(float, tag);
            ((f32x4*)Y.tslab + (tile0 + w) * 128)[lane] = (f32x4){v[0], v[1], v[2], tf};
            __builtin_amdgcn_s_sleep(1);
            if (!poll1(Y, par, ct, w, lane, tag, zs)) break;
        } else {
            slab_put_tagged(Y.tslab, tile0 + w, lane, v, tag);
            if (!slab_poll_sum_tagged(Y, par, ct, w, lane, tag, zs)) break;
        }
        acc += zs;
        __syncthreads();
    }
    if (acc[0] == -1.f) sink[0] = acc[1];
}
template <int ONE> static void run(PersistSync Y, float* sink, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    hipLaunchKernelGGL(k<ONE>, dim3(R * C), dim3(448), 0, 0, Y, sink, 200);   // warm
    Y.seq_base += 100000;
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<ONE>, dim3(R * C), dim3(448), 0, 0, Y, sink, iters);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.3f us per exchange\n", name, 1e3 * ms / iters);
}
int main() {
    float* tslab; unsigned *abortf, *xcc; float* sink;
    const size_t bytes = (size_t)2 * C * R * HT * 128 * 16;
    hipMalloc(&tslab, bytes); hipMalloc(&abortf, 8); hipMalloc(&xcc, R * C * 4); hipMalloc(&sink, 4);
    hipMemset(tslab, 0, bytes); hipMemset(abortf, 0, 8);
    PersistSync Y{tslab, abortf, xcc, 0, 100000};
    run<0>(Y, sink, "two entries per lane (tags)");
    Y.seq_base = 1000000; hipMemset(tslab, 0, bytes);
    run<1>(Y, sink, "one entry per lane");
    Y.seq_base = 2000000; hipMemset(tslab, 0, bytes);
    run<0>(Y, sink, "two entries per lane (tags)");
    Y.seq_base = 3000000; hipMemset(tslab, 0, bytes);
    run<1>(Y, sink, "one entry per lane");
    Y.seq_base = 4000000; hipMemset(tslab, 0, bytes);
    run<2>(Y, sink, "two written, one polled");
    Y.seq_base = 6000000; hipMemset(tslab, 0, bytes);
    run<4>(Y, sink, "one entry, packed tiles");
    Y.seq_base = 7000000; hipMemset(tslab, 0, bytes);
    run<5>(Y, sink, "one entry, indicator first");
    Y.seq_base = 5000000; hipMemset(tslab, 0, bytes);
    run<3>(Y, sink, "one entry, s_sleep before poll");
    return 0;
}
