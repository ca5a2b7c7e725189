// Probe: lane/register layouts of the f32 MFMA forms used by the step kernel.
// Exact small-integer data; prints the (block,row,col) -> (lane,reg) maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_4x4x1(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void probe_16x16x4(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
// timing: chains of independent accumulators
template <int NACC>
__global__ void time_4x4x1(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[100000] = (float)(t1 - t0);
}
template <int NACC>
__global__ void time_16x16x4(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[100000] = (float)(t1 - t0);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
    float *a, *b, *d;
    CK(hipMalloc(&a, 64 * 4)); CK(hipMalloc(&b, 64 * 4)); CK(hipMalloc(&d, 256 * 4));
    std::vector<float> ha(64), hb(64), hd(256);
    // 4x4x1 16 blocks: find which A lane and B lane feed D[lane][reg]
    // Use a = 1 on lane la only, b = 1 on lane lb only => D nonzero where block/row/col match.
    printf("== 4x4x1_16B: for each (A lane la, B lane lb in same block) list D positions\n");
    int amap_i[64], amap_b[64];
    for (int la = 0; la < 64; ++la) {
        // all B lanes = 1: D[b][i][*] = A_b[i]  -> shows which block/row A lane la is
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = 1.f; }
        CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice));
        probe_4x4x1<<<1, 64>>>(a, b, d); CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("A lane %2d -> D(lane,reg):", la);
        for (int i = 0; i < 256; ++i) if (hd[i] != 0) printf(" (%d,%d)", i / 4, i % 4);
        printf("\n");
    }
    for (int lb = 0; lb < 64; ++lb) {
        for (int i = 0; i < 64; ++i) { hb[i] = (i == lb); ha[i] = 1.f; }
        CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice));
        probe_4x4x1<<<1, 64>>>(a, b, d); CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("B lane %2d -> D(lane,reg):", lb);
        for (int i = 0; i < 256; ++i) if (hd[i] != 0) printf(" (%d,%d)", i / 4, i % 4);
        printf("\n");
    }
    printf("== 16x16x4: A lane -> D positions (B all ones), first 20 lanes + lane 16,32,48\n");
    int lanes[] = {0, 1, 2, 15, 16, 17, 32, 48, 63};
    for (int la : lanes) {
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = 1.f; }
        CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice));
        probe_16x16x4<<<1, 64>>>(a, b, d); CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("A lane %2d -> D(lane,reg):", la);
        int cnt = 0;
        for (int i = 0; i < 256; ++i) if (hd[i] != 0 && cnt++ < 6) printf(" (%d,%d)", i / 4, i % 4);
        printf(" ... n=%d\n", cnt);
    }
    for (int lb : lanes) {
        for (int i = 0; i < 64; ++i) { hb[i] = (i == lb); ha[i] = 1.f; }
        CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice));
        probe_16x16x4<<<1, 64>>>(a, b, d); CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
        printf("B lane %2d -> D(lane,reg):", lb);
        int cnt = 0;
        for (int i = 0; i < 256; ++i) if (hd[i] != 0 && cnt++ < 6) printf(" (%d,%d)", i / 4, i % 4);
        printf(" ... n=%d\n", cnt);
    }
    // timing
    float* out; CK(hipMalloc(&out, 100001 * 4 + 1024));
    int iters = 4096;
    std::vector<float> ho(1);
#define TIME(K, N) { K<N><<<1, 64>>>(out, iters); CK(hipDeviceSynchronize()); K<N><<<1, 64>>>(out, iters); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(ho.data(), out + 100000, 4, hipMemcpyDeviceToHost)); printf(#K " nacc=%d: %.2f cycles(clock64 ticks)/mfma\n", N, ho[0] / (iters * (double)N)); }
    TIME(time_4x4x1, 1) TIME(time_4x4x1, 2) TIME(time_4x4x1, 4) TIME(time_4x4x1, 8)
    TIME(time_16x16x4, 1) TIME(time_16x16x4, 2) TIME(time_16x16x4, 4)
    // two waves per SIMD: 512 threads
    {
        time_4x4x1<4><<<1, 512>>>(out, iters); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), out + 100000, 4, hipMemcpyDeviceToHost)); printf("time_4x4x1 nacc=4, 8 waves/WG: %.2f ticks/mfma (per wave)\n", ho[0] / (iters * 4.0));
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz wallclock rate=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.clockInstructionRate);
    return 0;
}
