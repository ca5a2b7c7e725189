// rnde_x3.h -- the Dense-layer GEMMs of the stage engine on the MATRIX CORES: exact three-way bf16 splitting of fp32 operands.
//
// Why (round 6, tools/micro/coexec.hip + valu_rate.hip, profiles/r06_coexec_micro.csv): on gfx950 the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate, 64 FLOP/clk/SIMD, and does not overlap with the vector instructions of any wave on its
// SIMD -- a wave issuing it back to back and a partner wave issuing v_fma_f32 take the SUM of their times (312 k cycles = 132 k + 189 k), whatever the
// priorities; the same pair with v_mfma_f32_16x16x32_bf16 takes the MAXIMUM (191 k = max(73 k, 189 k)) and SQ_VALU_MFMA_COEXEC_CYCLES counts it.  The
// fp32 "matrix" instruction is executed by the vector ALUs; the matrix cores proper take 16-bit (or narrower) inputs at 16x the rate.
//
// An fp32 number is EXACTLY the sum of three bf16 numbers (24 significant bits = 8 + 8 + 8):  x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh),
// xl = x - xh - xm (rounding to nearest at each level leaves a remainder of at most 16, then 8 significant bits: representable).  A product of two
// such sums has nine exact bf16 x bf16 terms; the six of weight >= 2^-16 relative to the leading one (hh, hm, mh, mm, hl, lh) carry the fp32 product
// to 2^-24: the same accuracy class as an fp32 multiply.  Each term is one v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, products exact in
// fp32, accumulation in fp32).  Accumulated smallest terms first, a K = 784 dot product of Glorot weights against [0, 1) data comes out 3.3e-7 from
// fp64 where the fp32 MFMA's k-ordered FMA chain is 1.1e-6 (largest error over the largest result; rms 1.5e-7 against 5.2e-7; numpy model of this scheme, tools/x3_model.py): the emulation is MORE accurate than
// the instruction it replaces, because 25 block accumulations replace 784 roundings.
// Cost per 32 k-values of a 16 x 16 tile: 6 matrix-core instructions of 16 cycles = 96 cycles, against 8 fp32 MFMAs of 32 cycles = 256 on the
// vector ALUs -- and the element-wise work (tanh, stage combinations, the splitting itself) now runs BESIDE them.
#pragma once
#include <hip/hip_runtime.h>

namespace rnde {

typedef unsigned x3u4 __attribute__((ext_vector_type(4)));       // one A / B fragment of v_mfma_f32_16x16x32_bf16: 8 bf16 per lane
typedef unsigned x3u2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3bf8 __attribute__((ext_vector_type(8)));
typedef float x3f4 __attribute__((ext_vector_type(4)));

constexpr int kX3K = 136;                                       // bf16 per (plane, column) row of an LDS operand image: 128 k-values + 8 of padding (16-lane b128 reads hit 64 different banks)
constexpr int kX3PlaneShorts = 16 * kX3K;                       // one plane of a 16-column operand
constexpr int kX3ImageFloats = 3 * kX3PlaneShorts / 2;          // three planes, in floats (LDS is carved in floats)

// two fp32 -> packed bf16 pair (low half = first), round to nearest even: v_cvt_pk_bf16_f32 (gfx950)
__device__ __forceinline__ unsigned x3_cvt2(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x0, x1) -> the three packed planes; x == hi + mid + lo exactly, component by component.
// The remainder x - float(hi) is representable, so ANY correctly rounded way of forming it gives the same bits.  RNDE_X3_DOT2 = 1 forms it with
// v_dot2_f32_bf16: D = A.lo * B.lo + A.hi * B.hi + C with B = {-1, 0} / {0, -1} and C = x -- two instructions per pair and level where rebuilding float(hi) from the
// packed pair (shift / mask) and subtracting takes four.  tools/micro/dot2_exact.hip: 4.29e9 random pairs (every exponent, subnormals, ties of the bf16 rounding), the
// three planes bit-equal in all but the 5e-4 of them whose hi overflows to Inf (|x| > 3.39e38, where the subtraction form is garbage too).  Two traps met on the way,
// both in that probe's history: (1) as inline assembly the pair v_cvt_pk_bf16_f32 -> v_dot2 read a STALE register (the compiler does not know an asm's hazards): both are
// compiler builtins here; (2) with B as a compile-time constant the compiler emits `v_dot2c_f32_bf16 v, -1.0, v`, an inline constant this hardware reads as an f16
// pattern: the two constants are kept in registers the optimiser cannot see through (one empty asm each, not volatile: it may be hoisted and merged).
// MEASURED AND NOT ADOPTED (round 6): exact, but no faster -- forward attempt 20.65 against 20.72-20.82 us, reversed attempt 24.79 against 24.85, and the weight-gradient
// kernels, whose bound IS the splitting's issue time, slower (rev_rest 0.330 against 0.317 ms): a v_dot2c costs more issue time than the two instructions it replaces.
#ifndef RNDE_X3_DOT2
#define RNDE_X3_DOT2 0
#endif
typedef __bf16 x3bf2 __attribute__((ext_vector_type(2)));
typedef float x3f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#pragma clang fp contract(off)
#if RNDE_X3_DOT2
    unsigned kl = 0x0000BF80u, kh = 0xBF800000u;      // {-1, 0}, {0, -1} as packed bf16 pairs
    asm("" : "+s"(kl));
    asm("" : "+s"(kh));
    const x3bf2 bl = __builtin_bit_cast(x3bf2, kl), bh = __builtin_bit_cast(x3bf2, kh);
    const x3bf2 h = __builtin_convertvector((x3f2){x0, x1}, x3bf2);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, bl, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(h, bh, x1, false);
    const x3bf2 m = __builtin_convertvector((x3f2){r0, r1}, x3bf2);
    const float s0 = __builtin_amdgcn_fdot2_f32_bf16(m, bl, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(m, bh, r1, false);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((x3f2){s0, s1}, x3bf2));
#else
    hi = x3_cvt2(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = x3_cvt2(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
    lo = x3_cvt2(s0, s1);
#endif
}
// four consecutive k-values of one column (k0 .. k0 + 3, k0 a multiple of 4) into the three planes of an LDS operand image [plane][column][kX3K]
__device__ __forceinline__ void x3_store4(unsigned short* img, int col, int k0, const x3f4& v) {
    unsigned h01, m01, l01, h23, m23, l23;
    x3_split2(v[0], v[1], h01, m01, l01);
    x3_split2(v[2], v[3], h23, m23, l23);
    unsigned short* p = img + col * kX3K + k0;
    *(x3u2*)p = (x3u2){h01, h23};
    *(x3u2*)(p + kX3PlaneShorts) = (x3u2){m01, m23};
    *(x3u2*)(p + 2 * kX3PlaneShorts) = (x3u2){l01, l23};
}
// B fragment of k-step s (k = 32 s + 8 (lane >> 4) + j, j < 8) of plane pl, column lane & 15
__device__ __forceinline__ x3u4 x3_frag(const unsigned short* img, int pl, int s, int lane) {
    return *(const x3u4*)(img + pl * kX3PlaneShorts + (lane & 15) * kX3K + 32 * s + 8 * (lane >> 4));
}
__device__ __forceinline__ x3f4 x3_mfma(const x3u4& a, const x3u4& b, const x3f4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3bf8, a), __builtin_bit_cast(x3bf8, b), c, 0, 0, 0);
}
// one 16 x 16 output tile over NS k-steps of 32: A planes wa[s][plane] (registers), B planes from the LDS image.  Six terms per step; four
// accumulators, one or two magnitude classes each (2^-16: lh + hl | mm; 2^-8: mh + hm; 1: hh), added smallest first at the end.
// The B fragments of step s + 1 are requested BEFORE the six matrix instructions of step s are issued (sched_barrier pins that order): with the reads
// of a step directly in front of its instructions a phase was 12 exposed LDS round trips long (~1.5 k cycles for 0.4 k of matrix work; the first
// build of this kernel, profiles/r06_x3_stamps.txt).  Two instructions on one accumulator are at least two apart (the dependent-result latency of a
// 16-cycle instruction is more than one issue slot).
template <int NS>
__device__ __forceinline__ x3f4 x3_tile(const x3u4 (&wa)[NS][3], const unsigned short* img, int lane) {
    x3f4 aL = {0.f, 0.f, 0.f, 0.f}, aN = {0.f, 0.f, 0.f, 0.f}, aM = {0.f, 0.f, 0.f, 0.f}, aH = {0.f, 0.f, 0.f, 0.f};
    x3u4 bh = x3_frag(img, 0, 0, lane), bm = x3_frag(img, 1, 0, lane), bl = x3_frag(img, 2, 0, lane);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        x3u4 nh = bh, nm = bm, nl = bl;
        if (s + 1 < NS) { nh = x3_frag(img, 0, s + 1, lane); nm = x3_frag(img, 1, s + 1, lane); nl = x3_frag(img, 2, s + 1, lane); }
        __builtin_amdgcn_sched_barrier(0);
        aL = x3_mfma(wa[s][2], bh, aL);      // lo * hi
        aM = x3_mfma(wa[s][1], bh, aM);      // mid * hi
        aH = x3_mfma(wa[s][0], bh, aH);      // hi * hi
        aN = x3_mfma(wa[s][1], bm, aN);      // mid * mid
        aL = x3_mfma(wa[s][0], bl, aL);      // hi * lo
        aM = x3_mfma(wa[s][0], bm, aM);      // hi * mid
        __builtin_amdgcn_sched_barrier(0);
        bh = nh; bm = nm; bl = nl;
    }
    return ((aL + aN) + aM) + aH;
}

// ---- packed A operands (weights), split once per forward: image [tile][k-step s < 4][plane < 3][64 lanes] of 16-byte fragments, lane l holding
// A[16 tile + (l & 15)][k = 32 s + 8 (l >> 4) + j], j < 8.  Four images, the x3 counterparts of stage_pack_elem's (rnde_stage.h):
//   0 forward  layer 2 (phase B): tile = row tile T < MT of the D outputs; k < H: W2[row][k]; k == H: the time column; k == H + 1: b2[row]; else 0
//   1 forward  layer 1 (phase D): tile = hidden tile w * R + row block rb; row m = 16 w + (l & 15) < H; k = row inside the block: state row
//                                 16 WT rb + k, k < 16 WT (= 112); else 0
//   2 reverse  W1x^T   (phase B): tile = row tile T; m = state row; k < H: W1[k][m]; else 0
//   3 reverse  W2xt^T  (phase D): tile = hidden tile w * R + row block rb; m = 16 w + (l & 15) <= H (m == H: the time column); k as in image 1: W2[state row][m]
// Parameter layout as stage_pack_elem: Flux.destructure order, W stored out x in column-major.  dst[0..1] forward, dst[2..3] reverse (nullptr: skipped).
struct X3PackDst { x3u4* d[4]; };
static __global__ void rnde_x3_pack_kernel(const float* __restrict__ p, const X3PackDst dst, int D, int H, int MT, int WT, int R, int HT) {
    const float* W1 = p;
    const float* b1 = W1 + (size_t)H * (D + 1);
    const float* W2 = b1 + H;
    const float* b2 = W2 + (size_t)D * (H + 1);
    const long long nB = (long long)MT * 4 * 64, nD = (long long)HT * R * 4 * 64, per = nB + nD;
    const int nimg = dst.d[2] ? 2 : 1;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nimg * per; i += (long long)gridDim.x * blockDim.x) {
        const int rev = (int)(i / per);
        const long long ii = i - rev * per;
        const bool isD = ii >= nB;
        const long long q = isD ? ii - nB : ii;
        const int l = (int)(q & 63), s = (int)((q >> 6) & 3), tile = (int)(q >> 8);
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + 8 * (l >> 4) + j;
            float w = 0.f;
            if (!isD) {
                const int m = 16 * tile + (l & 15);
                if (m < D) {
                    if (!rev) w = k <= H ? W2[(size_t)k * D + m] : (k == H + 1 ? b2[m] : 0.f);
                    else if (k < H) w = W1[(size_t)m * H + k];
                }
            } else {
                const int wt = tile / R, rb = tile % R;
                const int m = 16 * wt + (l & 15), row = 16 * WT * rb + k;
                if (k < 16 * WT && row < D) {
                    if (!rev) { if (m < H) w = W1[(size_t)row * H + m]; }
                    else if (m <= H) w = W2[(size_t)m * D + row];
                }
            }
            wv[j] = w;
        }
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x3_split2(wv[2 * j], wv[2 * j + 1], hi[j], mid[j], lo[j]);
        x3u4* o = dst.d[2 * rev + (isD ? 1 : 0)] + ((size_t)(tile * 4 + s) * 3) * 64 + l;
        o[0] = (x3u4){hi[0], hi[1], hi[2], hi[3]};
        o[64] = (x3u4){mid[0], mid[1], mid[2], mid[3]};
        o[128] = (x3u4){lo[0], lo[1], lo[2], lo[3]};
    }
}

}  // namespace rnde
