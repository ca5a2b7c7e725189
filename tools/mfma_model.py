"""Fit candidate arithmetic models to the raw fp32 MFMA results dumped by tools/micro/mfma_numerics.hip (run on the MI355X).

Used by the oracle's device-order summation mode (oracle/rnde_oracle.c, `orc_set_sum_order`): the CPU restatement has to add up
a K = 4 matrix instruction the way the matrix pipe does.  Exact rational arithmetic on the host (fractions), one rounding
function, and a handful of candidate orders; prints the fraction of outputs each model reproduces bit for bit.

    python tools/mfma_model.py gpurun_out/r03/mfma_numerics.bin
"""
import struct
import sys
from fractions import Fraction

import numpy as np


def rn(fr):
    """round a Fraction to the nearest fp32 (ties to even), returned as np.float32"""
    if fr == 0:
        return np.float32(0.0)
    # float(Fraction) is correctly rounded to double; double -> float32 double rounding is the hazard, so do it exactly
    s = -1 if fr < 0 else 1
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    # a in [2^e, 2^(e+1)); fp32 spacing 2^(e-23) (ignore subnormals: not exercised)
    ulp = Fraction(2) ** (e - 23)
    q = a / ulp
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(s * float(n * ulp))


def F(x):
    return Fraction(float(x))


def models(a, b, c):
    """a[k], b[k] (k = 0..K-1), c: exact Fractions.  Returns {name: fp32}"""
    K = len(a)
    out = {}
    # sequential FMA chains
    for name, order in (("fma_seq_0..K", range(K)), ("fma_seq_K..0", range(K - 1, -1, -1))):
        d = c
        for k in order:
            d = F(rn(d + a[k] * b[k]))
        out[name] = rn(d)
    # fused: exact dot + c, one rounding
    out["fused_once"] = rn(c + sum(a[k] * b[k] for k in range(K)))
    # products rounded to fp32 first, then added sequentially to c
    d = c
    for k in range(K):
        d = F(rn(d + F(rn(a[k] * b[k]))))
    out["mul_rounded_seq"] = rn(d)
    # exact dot rounded, then added to c
    out["dot_rounded_then_c"] = rn(c + F(rn(sum(a[k] * b[k] for k in range(K)))))
    if K == 4:
        # pairwise: (k0,k1) and (k2,k3) exact pairs rounded, added to c in order
        p01 = F(rn(a[0] * b[0] + a[1] * b[1]))
        p23 = F(rn(a[2] * b[2] + a[3] * b[3]))
        out["pairs_then_c"] = rn(F(rn(c + p01)) + p23)
        # c + pair01 fused, then + pair23 fused
        out["c_pair01_pair23"] = rn(F(rn(c + a[0] * b[0] + a[1] * b[1])) + a[2] * b[2] + a[3] * b[3])
        out["c_pair02_pair13"] = rn(F(rn(c + a[0] * b[0] + a[2] * b[2])) + a[1] * b[1] + a[3] * b[3])
    if K == 2:
        out["c_plus_pair"] = rn(c + a[0] * b[0] + a[1] * b[1])
    return out


def main(path, max_cases=24, max_out=64):
    raw = open(path, "rb").read()
    magic, ncase, nexp = struct.unpack_from("<III", raw, 0)
    assert magic == 0x314E464D
    off = 12
    w = np.frombuffer(raw, dtype=np.float32, offset=off)
    pos = 0

    def take(n):
        nonlocal pos
        v = w[pos:pos + n]
        pos += n
        return v

    score = {}
    total = 0
    first_bad = {}
    for n in range(ncase):
        a = take(64); b = take(64); c = take(256); d = take(256)
        if n >= max_cases:
            continue
        # 16x16x4: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D lane l reg r = D[4*(l>>4)+r][l&15]
        for l in range(0, 64, max(1, 64 // (max_out // 4))):
            for r in range(4):
                i, j = 4 * (l >> 4) + r, l & 15
                av = [F(a[16 * k + i]) for k in range(4)]
                bv = [F(b[16 * k + j]) for k in range(4)]
                m = models(av, bv, F(c[l * 4 + r]))
                total += 1
                for name, v in m.items():
                    ok = v.tobytes() == np.float32(d[l * 4 + r]).tobytes()
                    score[name] = score.get(name, 0) + int(ok)
                    if not ok and name not in first_bad:
                        first_bad[name] = (n, l, r, float(v), float(d[l * 4 + r]))
    print(f"v_mfma_f32_16x16x4_f32: {total} outputs checked")
    for name, s in sorted(score.items(), key=lambda kv: -kv[1]):
        print(f"  {name:22s} {s / total:8.4f}   first mismatch {first_bad.get(name)}")
    # 32x32x2: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31], D lane l reg r = D[8*(r>>2) + 4*(l>>5) + (r&3)][l&31]
    score, total, first_bad = {}, 0, {}
    for n in range(ncase):
        a = take(64); b = take(64); c = take(1024); d = take(1024)
        if n >= max_cases:
            continue
        for l in range(0, 64, 8):
            for r in range(0, 16, 3):
                i, j = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), l & 31
                av = [F(a[32 * k + i]) for k in range(2)]
                bv = [F(b[32 * k + j]) for k in range(2)]
                m = models(av, bv, F(c[l * 16 + r]))
                total += 1
                for name, v in m.items():
                    ok = v.tobytes() == np.float32(d[l * 16 + r]).tobytes()
                    score[name] = score.get(name, 0) + int(ok)
                    if not ok and name not in first_bad:
                        first_bad[name] = (n, l, r, float(v), float(d[l * 16 + r]))
    print(f"v_mfma_f32_32x32x2_f32: {total} outputs checked")
    for name, s in sorted(score.items(), key=lambda kv: -kv[1]):
        print(f"  {name:22s} {s / total:8.4f}   first mismatch {first_bad.get(name)}")
    # transcendental samples: error of the hardware instructions / tanh_fast in ulps against correctly rounded values
    xe = take(2 * nexp).reshape(-1, 2); xr = take(2 * nexp).reshape(-1, 2); xt = take(2 * nexp).reshape(-1, 2)

    def ulps(got, ref64):
        ref32 = ref64.astype(np.float32)
        u = np.spacing(np.abs(ref32)).astype(np.float64)
        return (got.astype(np.float64) - ref64) / u, (got == ref32).mean()
    e, eq = ulps(xe[:, 1], np.exp2(xe[:, 0].astype(np.float64)))
    print(f"v_exp_f32: max |err| {np.abs(e).max():.3f} ulp, mean {e.mean():+.3f}, equal to the correctly rounded value {eq:.3f}")
    e, eq = ulps(xr[:, 1], 1.0 / xr[:, 0].astype(np.float64))
    print(f"v_rcp_f32: max |err| {np.abs(e).max():.3f} ulp, mean {e.mean():+.3f}, equal to the correctly rounded value {eq:.3f}")
    e, eq = ulps(xt[:, 1], np.tanh(xt[:, 0].astype(np.float64)))
    print(f"tanh_fast: max |err| {np.abs(e).max():.3f} ulp, rms {np.sqrt((e * e).mean()):.3f}, mean {e.mean():+.3f}, correctly rounded {eq:.3f}")
    np.save(path + ".tanh.npy", xt)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03/mfma_numerics.bin")
