"""numpy model of matrix mode 1 (csrc/rnde_x3.h): how far from fp64 is a K = 784 Dense product when every fp32 operand is split exactly into three
bf16 numbers and the six leading cross products are accumulated per 32-k block (each block: four exact 8-term sums added with one rounding each, the model
tools/micro/mfma_bf16_numerics.hip fits to the matrix core), against the fp32-input MFMA's k-ordered FMA chain (tools/mfma_model.py)?

    python tools/x3_model.py            # prints the relative errors (max: 3.3e-7 vs 1.1e-6, rms: 1.5e-7 vs 5.2e-7 on Glorot weights x [0, 1) data)

CPU only; documentation of the numerics, not part of the product or of the oracle (oracle/rnde_oracle.c restates the same scheme in C)."""
import numpy as np


def bf16(x):
    """round-to-nearest-even fp32 -> bf16, returned as fp32"""
    u = np.asarray(x, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16(x)
    r = (x - h).astype(np.float32)
    m = bf16(r)
    lo = (r - m).astype(np.float32)
    assert np.array_equal(bf16(lo), lo)      # the remainder is a bf16 number: the split is exact
    return h, m, lo


def mfma_bf16(a, b, c):
    """one 32-k instruction on the fitted model: a (M, 32), b (32, N), c (M, N) fp32"""
    for q in range(4):
        s = a[:, 8 * q:8 * q + 8].astype(np.float64) @ b[8 * q:8 * q + 8].astype(np.float64)      # 8 products of bf16 numbers: exact in fp64
        c = (c.astype(np.float64) + s).astype(np.float32)
    return c


def x3_product(A, Bm):
    """A (M, K) @ Bm (K, N) in matrix mode 1; K padded to a multiple of 32"""
    M, K = A.shape
    Kp = -(-K // 32) * 32
    A = np.pad(A, ((0, 0), (0, Kp - K)))
    Bm = np.pad(Bm, ((0, Kp - K), (0, 0)))
    ah, am, al = split3(A)
    bh, bm, bl = split3(Bm)
    aL = np.zeros((M, Bm.shape[1]), np.float32)
    aN, aM, aH = aL.copy(), aL.copy(), aL.copy()
    for s in range(0, Kp, 32):
        k = slice(s, s + 32)
        aL = mfma_bf16(al[:, k], bh[k], aL)
        aM = mfma_bf16(am[:, k], bh[k], aM)
        aH = mfma_bf16(ah[:, k], bh[k], aH)
        aN = mfma_bf16(am[:, k], bm[k], aN)
        aL = mfma_bf16(ah[:, k], bl[k], aL)
        aM = mfma_bf16(ah[:, k], bm[k], aM)
    return ((aL + aN) + aM) + aH


def fma_chain(A, Bm):
    """the fp32-input MFMA: one fused multiply-add per k, in k order"""
    acc = np.zeros((A.shape[0], Bm.shape[1]), np.float64)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k + 1].astype(np.float64) * Bm[k:k + 1].astype(np.float64)).astype(np.float32).astype(np.float64)      # fp32 x fp32 is exact in fp64: one rounding per step
    return acc.astype(np.float32)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    K, M, N = 784, 112, 64
    W = (rng.uniform(-1, 1, (M, K)) * np.sqrt(6.0 / (K + 100))).astype(np.float32)
    X = rng.uniform(0, 1, (K, N)).astype(np.float32)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    scale, rms = np.abs(ref).max(), np.sqrt(np.mean(ref ** 2))
    for name, v in (("matrix mode 1 (bf16 x 3, six terms)", x3_product(W, X)), ("matrix mode 0 (fp32 FMA chain)     ", fma_chain(W, X))):
        print(f"K = {K}  {name}: max |error| / max |result| {np.abs(v - ref).max() / scale:.2e}   rms error / rms result {np.sqrt(np.mean((v - ref) ** 2)) / rms:.2e}")
