// dot2_exact.hip -- can v_dot2_f32_bf16 form the remainders of the exact three-way bf16 split (csrc/rnde_x3.h: x = hi + mid + lo)?
// The split needs r = x - float(hi) EXACTLY (it is representable); today that is a shift / mask to rebuild float(hi) from the packed pair plus a subtraction:
// 4 vector instructions per pair and level.  v_dot2_f32_bf16 D = A.lo * B.lo + A.hi * B.hi + C with B = {-1, 0} / {0, -1} and C = x would be 2.  This probe runs
// both forms over random bit patterns (every exponent, subnormals included), ties of the bf16 rounding and small integers, and counts the pairs whose three
// packed planes differ.  0 differing = the instruction can replace the sequence bit for bit.        hipcc --offload-arch=gfx950 -O2 -o dot2_exact dot2_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// (compiler builtins, not inline assembly: the first build of this probe used asm for both instructions and 72 % of the pairs "differed" -- the compiler does not
//  know an asm's hazards and left no wait state between v_cvt_pk_bf16_f32 and the v_dot2 that reads its result: the dot read a stale register)
__device__ __forceinline__ unsigned cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, bf2)); }
__device__ __forceinline__ void split_sub(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#pragma clang fp contract(off)
    hi = cvt2(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = cvt2(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
    lo = cvt2(s0, s1);
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), c, false); }
__device__ __forceinline__ void split_dot(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    unsigned kLo = 0x0000BF80u, kHi = 0xBF800000u;      // {-1, 0} and {0, -1} as packed bf16 pairs -- in REGISTERS the compiler cannot see through: folded, it emits
    asm volatile("" : "+v"(kLo), "+v"(kHi));           // `v_dot2c_f32_bf16 v, -1.0, v` with an inline constant the hardware reads as an f16 pattern (second build: 99.97 % differed)
    hi = cvt2(x0, x1);
    const float r0 = dot2(hi, kLo, x0), r1 = dot2(hi, kHi, x1);
    mid = cvt2(r0, r1);
    const float s0 = dot2(mid, kLo, r0), s1 = dot2(mid, kHi, r1);
    lo = cvt2(s0, s1);
}
__device__ __forceinline__ unsigned rnd(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 32); }
__global__ void probe(unsigned long long n_per_thread, unsigned long long* bad, unsigned* example) {
    unsigned long long s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    unsigned long long nbad = 0;
    for (unsigned long long i = 0; i < n_per_thread; ++i) {
        unsigned u0 = rnd(s), u1 = rnd(s);
        const unsigned kind = rnd(s) & 7u;
        if (kind == 0) { u0 = (u0 & 0xFFFF0000u) | 0x8000u; u1 = (u1 & 0xFFFF0000u) | 0x7FFFu; }      // exact ties / just below a tie of the first rounding
        if (kind == 1) { u0 &= 0x807FFFFFu; u1 &= 0x80FFFFFFu; }                                           // subnormal / smallest normals
        if (kind == 2) { u0 = (u0 & 0x807FFFFFu) | 0x7F000000u; u1 = (u1 & 0x807FFFFFu) | 0x7E800000u; }  // the largest exponents
        if ((u0 & 0x7F800000u) == 0x7F800000u) u0 &= 0xBFFFFFFFu;                                         // no Inf / NaN inputs
        if ((u1 & 0x7F800000u) == 0x7F800000u) u1 &= 0xBFFFFFFFu;
        const float x0 = __uint_as_float(u0), x1 = __uint_as_float(u1);
        unsigned a[3], b[3];
        split_sub(x0, x1, a[0], a[1], a[2]);
        split_dot(x0, x1, b[0], b[1], b[2]);
        if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) {
            if (nbad == 0 && atomicAdd(&example[0], 1u) == 0u) { example[1] = u0; example[2] = u1; example[3] = a[1]; example[4] = b[1]; example[5] = a[2]; example[6] = b[2]; }
            ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    unsigned long long* bad; unsigned* ex;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&ex, 32)); CK(hipMemset(bad, 0, 8)); CK(hipMemset(ex, 0, 32));
    const unsigned long long per = 4096;
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, per, bad, ex);
    CK(hipDeviceSynchronize());
    unsigned long long hb; unsigned he[8];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(he, ex, 32, hipMemcpyDeviceToHost));
    printf("pairs tested %llu, pairs whose planes differ between the subtraction form and the v_dot2_f32_bf16 form: %llu\n", 4096ull * 256ull * per, hb);
    if (hb) printf("first example: x0 %08x x1 %08x  mid sub %08x dot %08x  lo sub %08x dot %08x\n", he[1], he[2], he[3], he[4], he[5], he[6]);
    return 0;
}
