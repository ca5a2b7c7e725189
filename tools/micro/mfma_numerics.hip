// mfma_numerics.hip -- what does one fp32 MFMA compute, bit for bit?  (Round 3: the oracle's device-order summation mode,
// oracle/rnde_oracle.c `sum_order`, has to accumulate a K = 4 MFMA the way the matrix pipe does.)
// Dumps raw results for random operands of mixed magnitude; tools/mfma_model.py fits the candidate models offline
// (sequential FMA chain, reversed, fused single rounding, products rounded first, pairwise ...).
// Also dumps v_exp_f32 / v_rcp_f32 / tanh_fast samples so the CPU restatement of tanh_fast can be checked the same way.
//
// file layout (little endian, all 4-byte words): header {magic 'MFN1', ncase, nexp}; then per case (16x16x4):
//   a[64] b[64] c[256] d[256]  (lane-major: a[l], b[l], c[l*4+r], d[l*4+r]); then 32x32x2: a[64] b[64] c[1024] d[1024];
//   then nexp x {x, exp2(x)} ; nexp x {x, rcp(x)} ; nexp x {x, tanh_fast(x)}
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k16(const float* a, const float* b, const float* c, float* d, int ncase) {
    const int l = threadIdx.x;
    for (int n = blockIdx.x; n < ncase; n += gridDim.x) {
        f32x4 acc;
        for (int r = 0; r < 4; ++r) acc[r] = c[(size_t)n * 256 + l * 4 + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(size_t)n * 64 + l], b[(size_t)n * 64 + l], acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) d[(size_t)n * 256 + l * 4 + r] = acc[r];
    }
}
__global__ void k32(const float* a, const float* b, const float* c, float* d, int ncase) {
    const int l = threadIdx.x;
    for (int n = blockIdx.x; n < ncase; n += gridDim.x) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c[(size_t)n * 1024 + l * 16 + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(size_t)n * 64 + l], b[(size_t)n * 64 + l], acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) d[(size_t)n * 1024 + l * 16 + r] = acc[r];
    }
}
// the library's tanh_fast (regneuralde.jl_amd/csrc/rnde_device.h), verbatim arithmetic
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x), x2 = x * x;
    float p = -0.00671552f;
    p = fmaf(p, x2, 0.02136713f);
    p = fmaf(p, x2, -0.05391917f);
    p = fmaf(p, x2, 0.13333165f);
    p = fmaf(p, x2, -0.33333332f);
    const float small = fmaf(x, x2 * p, x);
    constexpr float L = 2.8853900817779268f;
    constexpr float Llo = (float)(2.8853900817779268 - (double)L);
    const float yh = ax * L;
    const float yl = fmaf(ax, L, -yh) + ax * Llo;
    float e = __builtin_amdgcn_exp2f(yh);
    e = fmaf(e, yl * 0.6931471805599453f, e);
    const float dd = e + 1.0f;
    float r = __builtin_amdgcn_rcpf(dd);
    r = fmaf(fmaf(-dd, r, 1.0f), r, r);
    float big = fmaf(-2.0f, r, 1.0f);
    big = ax > 9.1f ? 1.0f : big;
    return ax < 0.55f ? small : copysignf(big, x);
}
__global__ void kfun(const float* xe, const float* xr, const float* xt, float* oe, float* orr, float* ot, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { oe[i] = __builtin_amdgcn_exp2f(xe[i]); orr[i] = __builtin_amdgcn_rcpf(xr[i]); ot[i] = tanh_fast(xt[i]); }
}

static uint64_t s_ = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return (uint32_t)(s_ >> 32); }
static float urand() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
// mantissa-rich value with a random exponent in [-e, e]
static float rval(int e) {
    const float m = 1.0f + urand();
    const int ex = (int)(rnd() % (2 * e + 1)) - e;
    return ((rnd() & 1) ? -m : m) * ldexpf(1.0f, ex);
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "mfma_numerics.bin";
    const int ncase = 96, nexp = 65536;
    std::vector<float> a16(ncase * 64), b16(ncase * 64), c16(ncase * 256), d16(ncase * 256);
    std::vector<float> a32(ncase * 64), b32(ncase * 64), c32(ncase * 1024), d32(ncase * 1024);
    for (int n = 0; n < ncase; ++n) {
        // a third of the cases with equal magnitudes (GEMM-like), a third with spread exponents, a third with c tiny / huge
        const int ea = (n % 3 == 0) ? 0 : 6, ec = (n % 3 == 2) ? 12 : ea;
        for (int i = 0; i < 64; ++i) { a16[n * 64 + i] = rval(ea); b16[n * 64 + i] = rval(ea); a32[n * 64 + i] = rval(ea); b32[n * 64 + i] = rval(ea); }
        for (int i = 0; i < 256; ++i) c16[n * 256 + i] = (n % 6 == 5) ? 0.f : rval(ec);
        for (int i = 0; i < 1024; ++i) c32[n * 1024 + i] = (n % 6 == 5) ? 0.f : rval(ec);
    }
    std::vector<float> xe(nexp), xr(nexp), xt(nexp), oe(nexp), orr(nexp), ot(nexp);
    for (int i = 0; i < nexp; ++i) {
        xe[i] = 26.0f * urand();                 // exp2 argument range of tanh_fast: 2 log2(e) |x|, |x| < 9.1
        xr[i] = 1.0f + ldexpf(1.0f + urand(), (int)(rnd() % 26));   // e + 1
        xt[i] = (i & 1 ? -1.f : 1.f) * (i < nexp / 2 ? 3.0f * urand() : 10.0f * urand());
    }
    float *da, *db, *dc, *dd;
    auto run = [&](std::vector<float>& A, std::vector<float>& B, std::vector<float>& C, std::vector<float>& Dv, bool big) {
        hipMalloc(&da, A.size() * 4); hipMalloc(&db, B.size() * 4); hipMalloc(&dc, C.size() * 4); hipMalloc(&dd, Dv.size() * 4);
        hipMemcpy(da, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        if (big) k32<<<ncase, 64>>>(da, db, dc, dd, ncase); else k16<<<ncase, 64>>>(da, db, dc, dd, ncase);
        hipMemcpy(Dv.data(), dd, Dv.size() * 4, hipMemcpyDeviceToHost);
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    };
    run(a16, b16, c16, d16, false);
    run(a32, b32, c32, d32, true);
    float *dxe, *dxr, *dxt, *doe, *dor, *dot;
    hipMalloc(&dxe, nexp * 4); hipMalloc(&dxr, nexp * 4); hipMalloc(&dxt, nexp * 4); hipMalloc(&doe, nexp * 4); hipMalloc(&dor, nexp * 4); hipMalloc(&dot, nexp * 4);
    hipMemcpy(dxe, xe.data(), nexp * 4, hipMemcpyHostToDevice); hipMemcpy(dxr, xr.data(), nexp * 4, hipMemcpyHostToDevice); hipMemcpy(dxt, xt.data(), nexp * 4, hipMemcpyHostToDevice);
    kfun<<<(nexp + 255) / 256, 256>>>(dxe, dxr, dxt, doe, dor, dot, nexp);
    hipMemcpy(oe.data(), doe, nexp * 4, hipMemcpyDeviceToHost); hipMemcpy(orr.data(), dor, nexp * 4, hipMemcpyDeviceToHost); hipMemcpy(ot.data(), dot, nexp * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "hip error\n"); return 1; }
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    const uint32_t hdr[3] = {0x314E464Du, (uint32_t)ncase, (uint32_t)nexp};
    fwrite(hdr, 4, 3, f);
    for (int n = 0; n < ncase; ++n) {
        fwrite(&a16[n * 64], 4, 64, f); fwrite(&b16[n * 64], 4, 64, f); fwrite(&c16[n * 256], 4, 256, f); fwrite(&d16[n * 256], 4, 256, f);
    }
    for (int n = 0; n < ncase; ++n) {
        fwrite(&a32[n * 64], 4, 64, f); fwrite(&b32[n * 64], 4, 64, f); fwrite(&c32[n * 1024], 4, 1024, f); fwrite(&d32[n * 1024], 4, 1024, f);
    }
    for (int i = 0; i < nexp; ++i) { fwrite(&xe[i], 4, 1, f); fwrite(&oe[i], 4, 1, f); }
    for (int i = 0; i < nexp; ++i) { fwrite(&xr[i], 4, 1, f); fwrite(&orr[i], 4, 1, f); }
    for (int i = 0; i < nexp; ++i) { fwrite(&xt[i], 4, 1, f); fwrite(&ot[i], 4, 1, f); }
    fclose(f);
    printf("wrote %s: %d cases per MFMA form, %d function samples\n", path, ncase, nexp);
    return 0;
}
