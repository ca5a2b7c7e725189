// rnde_head.h -- the caller of the hot path, fused: postode Dense(D, C) + logitcrossentropy
// and their reverse in two small launches, so a training step is [solve] -> [head] -> [reverse solve] without a
// tape library in between.
// Replaces, for the classifier: reference src/models/supervised_classification.jl:44-45 (postode),
// experiments/mnist_node.jl:135 (Flux.Losses.logitcrossentropy) and the Tracker reverse of both.
//   p3 = Flux.destructure(Dense(D, C)) = [vec(W) (C x D, column-major); b (C)]
//   logits = W u + b;  ce = mean_c( -sum_i y_ic * logsoftmax(logits)_ic );  delta = (softmax - y) / B
//   u-bar = W^T delta;  W-bar = delta u^T;  b-bar = rowsum(delta)
#pragma once
#include "rnde_device.h"

namespace rnde {

constexpr int kHeadMaxC = 16;

// one workgroup (4 waves) per batch column: every thread owns rows d = tid + 256 k, k < 4, and keeps its rows of W in registers
// for both products (logits = W u, u-bar = W^T delta); all loads of the first product are issued up front (the one-wave-per-column form it
// replaces walked 13 dependent iterations twice: 34 -> 25 us at B = 512; staging W through LDS was measured: no faster)
constexpr int kHeadRowsPerThread = 4;       // D <= 1024
// CC > 0: the number of classes as a compile-time constant (10 for the reference's classifier), CC = 0: any C <= kHeadMaxC
template <int CC>
__global__ __launch_bounds__(256) void rnde_head_col_kernel(const float* __restrict__ u, const float* __restrict__ p3,
                                                            const float* __restrict__ y, int D, int C, int B,
                                                            float* __restrict__ logits_out, float* __restrict__ ubar,
                                                            float* __restrict__ delta, float* __restrict__ ce_col) {
    constexpr int NC = CC ? CC : kHeadMaxC;
    if (CC) C = CC;
    __shared__ float red[4][kHeadMaxC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = blockIdx.x;
    const float* W = p3;
    const float* b = p3 + (size_t)C * D;
    const float* uc = u + (size_t)c * D;
    float wv[kHeadRowsPerThread][NC], uv[kHeadRowsPerThread];
#pragma unroll
    for (int k = 0; k < kHeadRowsPerThread; ++k) {
        const int d = tid + 256 * k;
        uv[k] = d < D ? uc[d] : 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) wv[k][i] = (i < C && d < D) ? W[(size_t)d * C + i] : 0.f;
    }
    float acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        acc[i] = 0.f;
#pragma unroll
        for (int k = 0; k < kHeadRowsPerThread; ++k) acc[i] = fmaf(wv[k][i], uv[k], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) if (i < C) { const float s = wave_sum_f(acc[i]); if (lane == 0) red[w][i] = s; }
    __syncthreads();
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (i < C) { acc[i] = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) + b[i]; mx = fmaxf(mx, acc[i]); }
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (i < C) se += expf(acc[i] - mx);
    const float lse = mx + logf(se);
    float ce = 0.f, dl[NC];
    const float invB = 1.f / (float)B;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        dl[i] = 0.f;
        if (i < C) {
            const float yv = y[(size_t)c * C + i];
            ce -= yv * (acc[i] - lse);
            dl[i] = (expf(acc[i] - lse) - yv) * invB;
        }
    }
    if (tid == 0) {
        ce_col[c] = ce;
        for (int i = 0; i < C; ++i) { delta[(size_t)c * C + i] = dl[i]; if (logits_out) logits_out[(size_t)c * C + i] = acc[i]; }
    }
#pragma unroll
    for (int k = 0; k < kHeadRowsPerThread; ++k) {
        const int d = tid + 256 * k;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) if (i < C) s = fmaf(wv[k][i], dl[i], s);
        if (d < D) ubar[(size_t)c * D + d] = s;
    }
}

// W-bar[i][d] = sum_c delta[i][c] u[d][c]: pass 1 -- one thread per (d, column chunk), partial[chunk][d*C + i]
constexpr int kHeadChunks = 32;
static __global__ __launch_bounds__(256) void rnde_head_wgrad_kernel(const float* __restrict__ u, const float* __restrict__ delta,
                                                              int D, int C, int B, float* __restrict__ partial) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y;
    const int per = (B + kHeadChunks - 1) / kHeadChunks;
    const int c0 = ch * per, c1 = min(B, c0 + per);
    if (d >= D) return;
    float acc[kHeadMaxC];
#pragma unroll
    for (int i = 0; i < kHeadMaxC; ++i) acc[i] = 0.f;
    // (eight columns' loads in flight at a time: one after the other, each of the chunk's 16 columns was a cold round trip -- 13 us for 8 MFLOP)
    for (int cb = c0; cb < c1; cb += 8) {
        float uv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) uv[j] = cb + j < c1 ? u[(size_t)(cb + j) * D + d] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j < c1 ? cb + j : c0;
#pragma unroll
            for (int i = 0; i < kHeadMaxC; ++i) if (i < C) acc[i] = fmaf(delta[(size_t)c * C + i], uv[j], acc[i]);
        }
    }
    float* o = partial + (size_t)ch * C * D;
    for (int i = 0; i < C; ++i) o[(size_t)d * C + i] = acc[i];
}
// pass 2 -- fixed-order sum of the chunk partials; bias gradient and mean cross entropy by one extra wave
static __global__ __launch_bounds__(256) void rnde_head_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ delta,
                                                               const float* __restrict__ ce_col, int D, int C, int B,
                                                               float* __restrict__ p3bar, float* __restrict__ ce_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C * D) {
        float v[kHeadChunks];
#pragma unroll
        for (int ch = 0; ch < kHeadChunks; ++ch) v[ch] = partial[(size_t)ch * C * D + i];     // (all 32 requested before the first add)
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < kHeadChunks; ++ch) s += v[ch];
        p3bar[i] = s;
    }
    // bias gradient (C sums over the columns) and the mean cross entropy: the LAST block's four waves share them (wave w: classes w, w + 4, ..;
    // wave 3 also the cross entropy) -- one wave doing the eleven reductions one after the other was the longest thing in this launch
    if (blockIdx.x == gridDim.x - 1) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int k = w; k < C; k += 4) {
            float s = 0.f;
            for (int c = lane; c < B; c += 64) s += delta[(size_t)c * C + k];
            s = wave_sum_f(s);
            if (lane == 0) p3bar[(size_t)C * D + k] = s;
        }
        if (w == 3) {
            float s = 0.f;
            for (int c = lane; c < B; c += 64) s += ce_col[c];
            s = wave_sum_f(s);
            if (lane == 0) *ce_out = s / (float)B;
        }
    }
}

// Optimiser(InvDecay(gamma), Momentum(eta, rho)) on one flat parameter group (include/rnde.h: rnde_momentum_step)
static __global__ __launch_bounds__(256) void rnde_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                            long long len, float inv_decay, float eta, float rho) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const float gs = g[i] * inv_decay;
    const float vn = rho * v[i] - eta * gs;
    v[i] = vn;
    p[i] = p[i] + vn;
}

// Flux.Optimise.ADAM(eta, (beta1, beta2)) on one flat parameter group (include/rnde.h: rnde_adam_step; the optimiser of
// experiments/mnist_nsde.jl).  bc1 = 1 - beta1^t, bc2 = 1 - beta2^t for the step being taken.
static __global__ __launch_bounds__(256) void rnde_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        long long len, float gscale, float eta, float b1, float b2, float bc1, float bc2, float eps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const float gs = g[i] * gscale;
    const float mn = b1 * m[i] + (1.f - b1) * gs;
    const float vn = b2 * v[i] + (1.f - b2) * gs * gs;
    m[i] = mn; v[i] = vn;
    p[i] = p[i] - mn / bc1 / (sqrtf(vn / bc2) + eps) * eta;
}

}  // namespace rnde
