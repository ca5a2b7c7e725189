// mfma_bf16_numerics.hip -- what does ONE v_mfma_f32_16x16x32_bf16 compute, bit for bit?  (Round 6: the oracle's mirror of matrix mode 1, csrc/rnde_x3.h,
// has to add a 32-term bf16 dot product to an fp32 accumulator the way the matrix core does.)
// Random operands of mixed magnitude (the three planes of split fp32 numbers differ by 2^8 and 2^16, accumulators are up to 2^20 larger than a term), the
// device result against candidate models evaluated on the host:
//   EXACT1   D = fl32( C + sum_k a_k b_k ) with the 32 products and the sum exact, ONE rounding to nearest even
//   EXACT1_TZ  the same, rounded toward zero
//   CHAIN    D = C; for k: D = fma32(a_k, b_k, D)                                  (what the fp32-input MFMA does)
//   HALVES   two exact 16-term sums, each added to C with its own rounding
//   QUADS    four exact 8-term sums (k = 8 q .. 8 q + 7: one lane group's share), added in q order with a rounding each
//   ALIGN_T  products aligned to the largest exponent of {C, products} and truncated to 2^-26.. of it before an exact sum (a narrow internal accumulator), rounded once
// Prints the bit-match rate of each model; tools/mfma_bf16_numerics.csv (argv[1]) gets the table.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D, int ncase) {      // A: [case][16 rows][32 k], B: [case][32 k][16 cols], C / D: [case][16][16]
    const int l = threadIdx.x, rc = l & 15, g = l >> 4;
    for (int n = blockIdx.x; n < ncase; n += gridDim.x) {
        uint16_t a[8], b[8];
        for (int j = 0; j < 8; ++j) { a[j] = A[((size_t)n * 16 + rc) * 32 + 8 * g + j]; b[j] = B[((size_t)n * 32 + 8 * g + j) * 16 + rc]; }
        u32x4 av = {(unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16), (unsigned)a[4] | ((unsigned)a[5] << 16), (unsigned)a[6] | ((unsigned)a[7] << 16)};
        u32x4 bv = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16), (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
        f32x4 acc;
        for (int r = 0; r < 4; ++r) acc[r] = C[((size_t)n * 16 + 4 * g + r) * 16 + rc];      // C/D: row = 4 (l >> 4) + r, col = l & 15
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[((size_t)n * 16 + 4 * g + r) * 16 + rc] = acc[r];
    }
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float rz32(long double v) {      // round toward zero to fp32
    float f = (float)v;
    if (fabsl((long double)f) > fabsl(v)) f = nextafterf(f, 0.f);
    return f;
}
int main(int argc, char** argv) {
    const int ncase = 512;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<uint16_t> A((size_t)ncase * 16 * 32), B((size_t)ncase * 32 * 16);
    std::vector<float> C((size_t)ncase * 256), D((size_t)ncase * 256);
    for (int n = 0; n < ncase; ++n) {
        const int kind = n % 4;      // 0: all O(1); 1: A is a "lo" plane (2^-16); 2: big accumulator; 3: mixed magnitudes per k
        for (int i = 0; i < 16 * 32; ++i) {
            float s = kind == 1 ? ldexpf(1.f, -16) : (kind == 3 ? ldexpf(1.f, -(int)(rng() % 12)) : 1.f);
            A[(size_t)n * 512 + i] = f2bf(U(rng) * s);
            B[(size_t)n * 512 + i] = f2bf(U(rng) * (kind == 3 ? ldexpf(1.f, -(int)(rng() % 12)) : 1.f));
        }
        for (int i = 0; i < 256; ++i) C[(size_t)n * 256 + i] = U(rng) * (kind == 2 ? 1024.f : (kind == 1 ? 1e-3f : 1.f)) * (n % 8 == 0 ? 0.f : 1.f);
    }
    uint16_t *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, ncase);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    // candidate models: a list of k-groups; each group's products are summed exactly, the group sums are added to the running value in list order with one fp32
    // rounding each (mode 0: to nearest even, mode 1: toward zero); "tree" variants add the group sums to each other first (exactly or rounded), then to C
    struct Model { const char* name; int ngroups; int gsize; int layout; int rz; int tree; };
    // layout 0: group q = k in [gsize q, gsize (q + 1)); layout 1 (two passes of a K = 16 instruction): pass h = 0, 1, lane group g = 0..3: k = 8 g + 4 h + j, j < 4, groups in (h, g) order;
    // layout 2: the same groups in (g, h) order
    const Model models[] = {
        {"EXACT1 (32 exact, one rounding)", 1, 32, 0, 0, 0}, {"EXACT1 toward zero", 1, 32, 0, 1, 0}, {"CHAIN (32 fma)", 32, 1, 0, 0, 0}, {"HALVES (2 x 16 exact)", 2, 16, 0, 0, 0},
        {"QUADS (4 x 8 exact, sequential)", 4, 8, 0, 0, 0}, {"QUADS toward zero", 4, 8, 0, 1, 0}, {"OCTS (8 x 4 exact, sequential)", 8, 4, 0, 0, 0},
        {"P2G4 (2 passes x 4 lane groups of 4)", 8, 4, 1, 0, 0}, {"G4P2 (4 lane groups x 2 passes of 4)", 8, 4, 2, 0, 0},
        {"P2x16 (2 passes, 16 exact each)", 2, 16, 1, 0, 0}, {"QUADS tree: exact quads, (q0+q1)+(q2+q3) rounded, + C", 4, 8, 0, 0, 1}, {"QUADS: quads summed exactly to fp32 first, then + C", 4, 8, 0, 0, 2},
    };
    const int nmodels = (int)(sizeof(models) / sizeof(models[0]));
    std::vector<long> match(nmodels, 0);
    long total = 0;
    double worst_exact = 0;
    auto rnd = [&](long double v, int rz) { return rz ? rz32(v) : (float)v; };
    for (int n = 0; n < ncase; ++n)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                const float c0 = C[(size_t)n * 256 + r * 16 + c], d = D[(size_t)n * 256 + r * 16 + c];
                long double p[32], sum = 0;
                for (int kk = 0; kk < 32; ++kk) { p[kk] = (long double)bf2f(A[((size_t)n * 16 + r) * 32 + kk]) * (long double)bf2f(B[((size_t)n * 32 + kk) * 16 + c]); sum += p[kk]; }
                for (int mi = 0; mi < nmodels; ++mi) {
                    const Model& M = models[mi];
                    long double gs[32];
                    for (int q = 0; q < M.ngroups; ++q) {
                        gs[q] = 0;
                        for (int j = 0; j < M.gsize; ++j) {
                            int k;
                            if (M.layout == 0) k = M.gsize * q + j;
                            else if (M.gsize == 4) { const int h = M.layout == 1 ? q / 4 : q % 2, g = M.layout == 1 ? q % 4 : q / 2; k = 8 * g + 4 * h + j; }
                            else { const int h = q; k = 8 * (j / 4) + 4 * h + (j % 4); }      // 16 per pass: lane groups g = j / 4
                            gs[q] += p[k];
                        }
                    }
                    float t;
                    if (M.tree == 0) { t = c0; for (int q = 0; q < M.ngroups; ++q) t = rnd((long double)t + gs[q], M.rz); }
                    else if (M.tree == 1) { const float a = (float)(gs[0] + gs[1]), b = (float)(gs[2] + gs[3]); t = (float)((long double)c0 + (long double)(float)((long double)a + (long double)b)); }
                    else { const float a = (float)(gs[0] + gs[1] + gs[2] + gs[3]); t = (float)((long double)c0 + (long double)a); }
                    match[mi] += memcmp(&t, &d, 4) == 0;
                }
                const double ex = (double)((long double)c0 + sum);
                if (ex != 0) worst_exact = fmax(worst_exact, fabs((double)d - ex) / fmax(fabs(ex), 1e-30));
                ++total;
            }
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "model,bit_equal,total,fraction\n");
    for (int i = 0; i < nmodels; ++i) {
        printf("%-60s %7ld / %ld bit-equal (%.2f %%)\n", models[i].name, match[i], total, 100.0 * match[i] / total);
        if (csv) fprintf(csv, "\"%s\",%ld,%ld,%.5f\n", models[i].name, match[i], total, (double)match[i] / total);
    }
    printf("largest |device - exact| / |exact| = %.3e (2^-24 = 5.96e-8)\n", worst_exact);
    if (csv) fclose(csv);
    return 0;
}
