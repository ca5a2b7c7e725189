"""Generates tests/golden/*.npz from the CPU oracle (fp32 and fp64 builds).

Run from the repo root:  python tests/golden/make_golden.py
These vectors pin (a) the oracle against regressions and (b) the HIP path on the GPU box, where
/root/reference and a Julia runtime do not exist.  They do NOT pin parity versus the Julia reference:
that reference ships no golden vectors and cannot be executed here (see oracle/rnde_oracle.h).
Inputs are regenerated in the tests from `lcg_uniform` (a portable LCG), so only outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, arch_latent, arch_mnist, arch_test_node  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg_uniform(n, seed, lo=0.0, hi=1.0):
    """Portable 64-bit LCG (Knuth MMIX) -> uniform doubles in [lo, hi)."""
    out = np.empty(n, dtype=np.float64)
    s = np.uint64(seed)
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for i in range(n):
            s = s * a + c
            out[i] = float(s >> np.uint64(11)) / float(1 << 53)
    return lo + (hi - lo) * out


def params_for(arch, seed, scale):
    parts = []
    for l in range(arch.n_layers):
        ine = arch.dims[l] + (1 if arch.time_dep else 0)
        o = arch.dims[l + 1]
        lim = scale * np.sqrt(6.0 / (ine + o))
        parts.append(lcg_uniform(ine * o, seed + 17 * l, -lim, lim))
        parts.append(lcg_uniform(o, seed + 17 * l + 5, -0.05, 0.05))
    return np.concatenate(parts)


CASES = {
    # name: (arch factory, B, tol, scale, t1, seed)
    "test_node_B1": (arch_test_node, 1, 1e-3, 3.0, 1.0, 11),       # the reference test's shape (test/test_node.jl:4-6)
    "test_node_B5": (arch_test_node, 5, 1e-3, 3.0, 1.0, 12),
    "mnist_small_B4": (lambda: arch_mnist(36, 10), 4, 1e-3, 4.0, 1.0, 13),
    "mnist_B3": (arch_mnist, 3, 1e-3, 3.0, 1.0, 14),
    "latent_B4": (arch_latent, 4, 1e-3, 2.0, 1.0, 15),             # chain engine (tests/test_gpu_golden.py)
}


def inputs(name):
    mk, B, tol, scale, t1, seed = CASES[name]
    arch = mk()
    p = params_for(arch, seed, scale)
    x = lcg_uniform(B * arch.dims[0], seed + 1000).reshape(B, arch.dims[0])
    wu = lcg_uniform(B * arch.dims[0], seed + 2000, -1.0, 1.0).reshape(B, arch.dims[0])
    return arch, p, x, wu, tol, t1


def main():
    for name in CASES:
        arch, p, x, wu, tol, t1 = inputs(name)
        out = {}
        for tag, dt in (("f32", np.float32), ("f64", np.float64)):
            o = Oracle(arch, dt, reltol=tol, abstol=tol, reg_kind=1)
            r = o.forward(x, p, 0.0, t1)
            assert r["rc"] == 0
            svbar = np.full(len(r["saveval"]), 25.0)
            xb, pb, tsb = o.backward(wu, svbar)
            # large parameter gradients are stored as a strided sample plus two checksums (keeps fixtures small)
            stride = 101 if pb.size > 20000 else 1
            out.update({f"u_{tag}": r["u"], f"nfe_{tag}": r["nfe"], f"saveval_{tag}": r["saveval"], f"steps_{tag}": r["steps"],
                        f"xbar_{tag}": xb, f"pbar_{tag}": pb[::stride], f"pbar_stride_{tag}": stride,
                        f"pbar_norm_{tag}": np.linalg.norm(pb.astype(np.float64)), f"pbar_sum_{tag}": pb.astype(np.float64).sum(),
                        f"tspanbar_{tag}": tsb})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "nfe", out["nfe_f32"], out["nfe_f64"], "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
