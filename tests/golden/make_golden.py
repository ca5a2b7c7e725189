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


# ---- second generation of fixtures (round 2): the stochastic layer and the tableau-as-data pair -----------------------------------
def lcg_normal(n, seed):
    """Standard normals from the portable LCG (Box-Muller on pairs of uniforms)."""
    u = lcg_uniform(2 * n, seed, 1e-12, 1.0)
    return np.sqrt(-2.0 * np.log(u[0::2])) * np.cos(2.0 * np.pi * u[1::2])


NSDE_CASES = {
    # name: (B, tol, drift scale, diffusion scale, controller, seed, n_pool): experiments/mnist_nsde.jl:73-80 shapes (D = 32, 32 -> 64 -> 32, 32 -> 32)
    "nsde_B8": (8, 0.14, 2.0, 0.5, {}, 31, 96),
    "nsde_B5_rejecting": (5, 0.1, 2.5, 0.8, dict(qmax=10.0, gamma=1.0, beta2=1e-9), 32, 400),   # oscillating controller: more than half rejected
}


def nsde_inputs(name):
    from oracle.oracle_sde import arch_nsde_diffusion, arch_nsde_drift
    B, tol, sf, sg, ctrl, seed, n_pool = NSDE_CASES[name]
    drift, diff = arch_nsde_drift(), arch_nsde_diffusion()
    p = np.concatenate([params_for(drift, seed, sf), params_for(diff, seed + 500, sg)])
    x = lcg_uniform(B * 32, seed + 1000, -1.0, 1.0).reshape(B, 32)
    wu = lcg_uniform(B * 32, seed + 2000, -1.0, 1.0).reshape(B, 32)
    noise = lcg_normal(n_pool * 2 * B * 32, seed + 3000).reshape(n_pool, 2, B, 32)
    return drift, diff, p, x, wu, noise, tol, ctrl


def dp5_inputs():
    """The latent-ODE dynamics integrated with Dormand-Prince 5(4) (RNDE_SOLVER_DP5, the tableau-as-data kernels)."""
    return inputs("latent_B4")


# ---- third generation (round 3): the natural run at the REFERENCE tolerance, summed in the device's order ---------------------------
DEVORDER_CASES = {
    # name: (B, tol, seed): MNIST shape, Glorot weights (scale 1), reltol = abstol = 1.4e-8 (experiments/mnist_node.jl:121-124)
    "mnist_B16_reftol_devorder": (16, 1.4e-8, 41),
}


def devorder_inputs(name):
    B, tol, seed = DEVORDER_CASES[name]
    arch = arch_mnist()
    p = params_for(arch, seed, 1.0)
    x = lcg_uniform(B * 784, seed + 1000).reshape(B, 784)
    return arch, p, x, tol


def main3():
    """At 1.4e-8 the fp32 step sequence is set by rounding noise, so only an oracle that accumulates as the device does
    (Oracle(sum_order=3): oracle/rnde_oracle.c orc_set_sum_order) yields a sequence the device can be held to.  Stored: the step log
    (t, dt, EEst, accepted), u_end, the saved values, and the sequential-k oracle's attempt count beside it."""
    for name in DEVORDER_CASES:
        arch, p, x, tol = devorder_inputs(name)
        o3 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96, sum_order=3)
        r = o3.forward(x, p)
        o0 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96).forward(x, p)
        o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96).forward(x, p)
        assert r["rc"] == 0 and o0["rc"] == 0 and o64["rc"] == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), u_devorder=r["u"], nfe_devorder=r["nfe"], steps_devorder=r["steps"],
                            saveval_devorder=r["saveval"], nfe_sequential=o0["nfe"], nfe_f64=o64["nfe"], u_f64=o64["u"])
        print(name, "attempts: device order", r["nattempts"], "sequential", o0["nattempts"], "fp64", o64["nattempts"],
              "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


def main5():
    """Round 6: the same natural run as the device's DEFAULT matrix mode computes it -- Oracle(sum_order=7), the mirror of csrc/rnde_x3.h (operands split exactly into
    three bf16 numbers, six cross products, each matrix instruction as four exact 8-term sums).  Stored beside the fp32-MFMA-order and sequential counts."""
    for name in DEVORDER_CASES:
        arch, p, x, tol = devorder_inputs(name)
        r = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96, sum_order=7).forward(x, p)
        o3 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96, sum_order=3).forward(x, p)
        o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1, max_attempts=96).forward(x, p)
        assert r["rc"] == 0
        out = name.replace("devorder", "x3")
        np.savez_compressed(os.path.join(HERE, out + ".npz"), u_x3=r["u"], nfe_x3=r["nfe"], steps_x3=r["steps"], saveval_x3=r["saveval"], nfe_devorder=o3["nfe"], u_f64=o64["u"])
        print(out, "attempts: matrix mode 1", r["nattempts"], "fp32-MFMA order", o3["nattempts"], "fp64", o64["nattempts"], "bytes", os.path.getsize(os.path.join(HERE, out + ".npz")))


def main2():
    from oracle.oracle_sde import SdeOracle
    for name in NSDE_CASES:
        drift, diff, p, x, wu, noise, tol, ctrl = nsde_inputs(name)
        out = {}
        for tag, dt in (("f32", np.float32), ("f64", np.float64)):
            o = SdeOracle(drift, diff, dt, tol, tol, tableau="SOSRI", max_attempts=399, **ctrl)
            r = o.forward(x, p, noise)
            assert r["rc"] == 0
            xb, pb = o.backward(wu, np.full(len(r["saveval"]), 5.0))
            out.update({f"u_{tag}": r["u"], f"nfe1_{tag}": r["nfe1"], f"nfe2_{tag}": r["nfe2"], f"saveval_{tag}": r["saveval"], f"steps_{tag}": r["steps"],
                        f"ndraws_{tag}": r["ndraws"], f"xbar_{tag}": xb, f"pbar_{tag}": pb})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "attempts", len(out["steps_f32"]), len(out["steps_f64"]), "rejected", int((out["steps_f32"][:, 3] == 0).sum()),
              "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))
    arch, p, x, wu, tol, t1 = dp5_inputs()
    out = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        o = Oracle(arch, dt, reltol=tol, abstol=tol, reg_kind=1, solver="DP5")
        r = o.forward(x, p, 0.0, t1)
        assert r["rc"] == 0
        xb, pb, tsb = o.backward(wu, np.full(len(r["saveval"]), 25.0))
        out.update({f"u_{tag}": r["u"], f"nfe_{tag}": r["nfe"], f"saveval_{tag}": r["saveval"], f"steps_{tag}": r["steps"], f"xbar_{tag}": xb,
                    f"pbar_{tag}": pb, f"tspanbar_{tag}": tsb})
    np.savez_compressed(os.path.join(HERE, "latent_dp5_B4.npz"), **out)
    print("latent_dp5_B4 nfe", out["nfe_f32"], out["nfe_f64"])


# round 5: the SDE layer's stiffness-estimate regulariser (reg_kind 2 on SOSRI2 = AutoSOSRI2(SOSRI2()), experiments/mnist_nsde.jl:51-61): same shapes and
# portable inputs as NSDE_CASES, saved values |eigen_est| / 10.6, cotangent 0.1 / n on them (lambda 0.1 x mean, :52, :99)
NSDE_STIFF_CASES = {"nsde_stiff_B8": (8, 0.14, 2.0, 0.5, {}, 41, 128), "nsde_stiff_B5_rejecting": (5, 0.1, 2.5, 0.8, dict(qmax=10.0, gamma=1.0, beta2=1e-9), 42, 400)}


def nsde_stiff_inputs(name):
    NSDE_CASES[name] = NSDE_STIFF_CASES[name]
    try:
        return nsde_inputs(name)
    finally:
        del NSDE_CASES[name]


def main4():
    from oracle.oracle_sde import SdeOracle
    for name in NSDE_STIFF_CASES:
        drift, diff, p, x, wu, noise, tol, ctrl = nsde_stiff_inputs(name)
        out = {}
        for tag, dt in (("f32", np.float32), ("f64", np.float64)):
            o = SdeOracle(drift, diff, dt, tol, tol, tableau="SOSRI2", reg_kind=2, max_attempts=399, **ctrl)
            r = o.forward(x, p, noise)
            assert r["rc"] == 0
            n = len(r["saveval"])
            xb, pb = o.backward(wu, np.full(n, 3.0))
            out.update({f"u_{tag}": r["u"], f"nfe1_{tag}": r["nfe1"], f"saveval_{tag}": r["saveval"], f"steps_{tag}": r["steps"], f"ndraws_{tag}": r["ndraws"],
                        f"norms_{tag}": o.eigen_norms(), f"xbar_{tag}": xb, f"pbar_{tag}": pb})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "attempts", len(out["steps_f32"]), len(out["steps_f64"]), "rejected", int((out["steps_f32"][:, 3] == 0).sum()), "saved", out["saveval_f64"][:4],
              "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stiff":          # only the round-5 fixtures
        main4()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "x3":            # only the round-6 fixture
        main5()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "devorder":      # only the round-3 fixture (the older ones stay byte-identical in git)
        main3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        main2()          # (the round-1 fixtures stay as committed)
    else:
        main()
