// ablation micro-benchmark of the chain engine's f evaluation (config 4 shapes): cycles per evaluation for one wave per SIMD
#include "../../regneuralde.jl_amd/csrc/rnde_chain.h"
#include <cstdio>
#include <vector>
using namespace rnde;
__global__ __launch_bounds__(256) void k(ChainGeo G, const float* frags, float* out, unsigned long long* st, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units = (G.nfrag_f + G.nfrag_b + 3) >> 2;
    chain_fill_lds(frags, smem, units, wave, lane);
    const float* FR = smem; const float* BF = FR + (size_t)G.nfrag_f * 64;
    float g[8], kv[8];
    for (int q = 0; q < 8; ++q) g[q] = 0.01f * (lane + q);
    unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        chain_eval<8>(G, FR, BF, 0.1f, g, kv, lane);
        for (int q = 0; q < 8; ++q) g[q] = 0.5f * g[q] + 0.1f * kv[q];
    }
    unsigned long long c1 = clock64();
    float s = 0; for (int q = 0; q < 8; ++q) s += g[q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) st[blockIdx.x] = c1 - c0;
}
int main() {
    ChainGeo G{}; G.n_layers = 8; G.time_dep = 0; G.pre_act = 1;
    int dims[9] = {20, 50, 20, 50, 20, 50, 20, 50, 20};
    int po = 0, fo = 0, bo = 0, to = 0;
    for (int l = 0; l <= 8; ++l) { G.width[l] = dims[l]; G.nks[l] = (dims[l] + 3) / 4; }
    for (int l = 0; l < 8; ++l) {
        G.act[l] = 1; G.poff[l] = po; po += dims[l] * dims[l + 1] + dims[l + 1];
        G.foff[l] = fo; fo += ((G.nks[l + 1] + 3) / 4) * G.nks[l];
        G.boff[l] = bo; bo += 4 * ((G.nks[l + 1] + 3) / 4);
        G.toff[l] = to; to += ((G.nks[l] + 3) / 4) * G.nks[l + 1];
    }
    G.nfrag_f = fo; G.nfrag_b = bo; G.nfrag_t = to; G.nksD = 5;
    std::vector<float> p(po); for (int i = 0; i < po; ++i) p[i] = 0.2f * ((i * 2654435761u >> 8) % 1000) / 1000.f - 0.1f;
    float *pd, *frags, *out; unsigned long long* st;
    hipMalloc(&pd, po * 4); hipMemcpy(pd, p.data(), po * 4, hipMemcpyHostToDevice);
    hipMalloc(&frags, (size_t)(fo + bo + to + 4) * 256); hipMalloc(&out, 64 * 256 * 4); hipMalloc(&st, 64 * 8);
    hipLaunchKernelGGL(rnde_chain_pack_kernel, dim3(64), dim3(256), 0, 0, pd, frags, G);
    const size_t lds = ((size_t)((fo + bo + 3) / 4) * 256 + 64) * 4;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(8), dim3(256), lds, 0, G, frags, out, st, 200);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("%s: %.0f cycles per f evaluation (%.2f us at 2.4 GHz); 184 MFMAs -> %.1f cycles per MFMA\n", VARIANT, h / 200.0, h / 200.0 / 2400.0, h / 200.0 / 184.0);
    return 0;
}
