"""Self-tests of the stochastic CPU oracle (oracle/rnde_sde_oracle.c): what pins it in the absence of Julia.

 1. the three tableaux (SOSRI, SOSRI2, SRIW1) satisfy Roessler's conditions for strong order 1.5;
 2. strong convergence order on geometric Brownian motion with the exact solution on the SAME path;
 3. the rejection-sampling-with-memory bookkeeping: adaptive runs with rejections walk one Brownian path
    (W(t1) has the right law, and the solution on it converges to the exact one);
 4. nfe1 = nfe2 = 2 + 4 * attempts; callback values EEst * dt; accepted steps tile [t0, t1];
 5. fp64 central finite differences of the reverse pass along a frozen step sequence.
"""
import numpy as np
import pytest

from oracle.oracle import make_arch
from oracle.oracle_sde import SdeOracle, arch_nsde_diffusion, arch_nsde_drift, nsde_params, sri_tableau


@pytest.mark.parametrize("name", ["SOSRI", "SOSRI2", "SRIW1"])
def test_tableau_strong_order_conditions(name):
    T = sri_tableau(name)
    e = np.ones(4)
    A0, A1, B0, B1 = T["A0"], T["A1"], T["B0"], T["B1"]
    al, b1, b2, b3, b4 = T["alpha"], T["beta1"], T["beta2"], T["beta3"], T["beta4"]
    tol = 3e-8 if name == "SOSRI2" else 1e-12   # SOSRI2's beta4 row comes out of a numerical optimisation (residual 1.4e-8)
    assert np.abs(A0 @ e - T["c0"]).max() < 1e-14 and np.abs(A1 @ e - T["c1"]).max() < 1e-14
    B1e, A1e, B0e = B1 @ e, A1 @ e, B0 @ e
    conds = [
        (al @ e, 1), (b1 @ e, 1), (b2 @ e, 0), (b3 @ e, 0), (b4 @ e, 0),
        (al @ B0e, 1), (al @ (A0 @ e), 0.5), (al @ B0e ** 2, 1.5),
        (b1 @ A1e, 1), (b1 @ B1e, 0), (b1 @ B1e ** 2, 1), (b1 @ (B1 @ B1e), 0), (b1 @ (A1 @ B0e), 0),
        (b2 @ A1e, 0), (b2 @ B1e, 1), (b2 @ B1e ** 2, 0), (b2 @ (B1 @ B1e), 0), (b2 @ (A1 @ B0e), 0),
        (b3 @ A1e, -1), (b3 @ B1e, 0), (b3 @ B1e ** 2, -1), (b3 @ (B1 @ B1e), 0), (b3 @ (A1 @ B0e), 0),
        (b4 @ A1e, 0), (b4 @ B1e, 0), (b4 @ B1e ** 2, 2), (b4 @ (B1 @ B1e), 1),
        (0.5 * b1 @ (A1 @ B0e) + b3 @ (A1 @ B0e) / 3, 0),
    ]
    for i, (v, want) in enumerate(conds):
        assert abs(v - want) < tol, (name, i, v, want)


def _gbm(a, b):
    """dX = a X dt + b X dW as Dense(1,1) drift and diffusion (identity activations, zero bias)."""
    drift = make_arch([1, 1], ["identity"], False)
    diff = make_arch([1, 1], ["identity"], False)
    p = np.array([a, 0.0, b, 0.0])
    return drift, diff, p


@pytest.mark.parametrize("name", ["SOSRI", "SOSRI2", "SRIW1"])
def test_strong_convergence_order_on_gbm(name):
    a, b = 1.2, 0.7
    drift, diff, p = _gbm(a, b)
    rng = np.random.default_rng(3)
    B = 4000
    x = np.ones((B, 1))
    errs, hs = [], []
    for nsteps in (8, 16, 32, 64):
        o = SdeOracle(drift, diff, np.float64, 1e9, 1e9, tableau=name, reg_kind=0, max_attempts=nsteps + 1)
        o.set_replay(np.full(nsteps, 1.0 / nsteps), np.ones(nsteps, np.int32))
        noise = rng.standard_normal((nsteps + 1, 2, B, 1))
        r = o.forward(x, p, noise)
        assert r["rc"] == 0 and r["nattempts"] == nsteps and r["ndraws"] == nsteps
        assert r["nfe1"] == 2 + 4 * nsteps and r["nfe2"] == 2 + 4 * nsteps
        w, _ = o.path_total(B)
        exact = np.exp((a - 0.5 * b * b) + b * w)
        errs.append(np.mean(np.abs(r["u"] - exact))); hs.append(1.0 / nsteps)
    slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
    assert slope > 1.3, (slope, errs)    # strong order 1.5 (1.0 would be Milstein, 0.5 Euler-Maruyama)


def test_adaptive_run_walks_one_brownian_path():
    """An aggressive controller (qmax = 10, gamma = 1, no PI memory; the recalled defaults hardly ever reject because
    qmax = 1.125 lets dt grow by 12.5 % per step only): two thirds of the attempts are rejected, and the rejections DEPEND on
    the noise (large increments fail).  Every increment used is a piece of one Brownian path, so
    (i) the solution converges to the exact GBM solution evaluated at the path's own W(t1), and (ii) W(t1) ~ N(0, t1 - t0):
    rejection does not bias the path (that is what the RSwM stacks are for)."""
    a, b = 0.5, 1.0
    drift, diff, p = _gbm(a, b)
    rng = np.random.default_rng(11)
    B = 1            # one scalar path per solve: a shared step-size controller over many columns would hide rejections' bias
    w_all, err_all, nrej = [], [], 0
    for rep in range(400):
        o = SdeOracle(drift, diff, np.float64, 1e-2, 1e-2, reg_kind=1, max_attempts=4000, qmax=10.0, gamma=1.0, beta2=1e-9)
        noise = rng.standard_normal((4001, 2, B, 1))
        r = o.forward(np.ones((B, 1)), p, noise)
        assert r["rc"] == 0
        st = r["steps"]
        nrej += int((st[:, 3] == 0).sum())
        acc = st[st[:, 3] == 1]
        assert abs(acc[:, 1].sum() - 1.0) < 1e-12 and np.allclose(acc[1:, 0], acc[:-1, 0] + acc[:-1, 1], atol=1e-14)
        assert r["nfe1"] == 2 + 4 * r["nattempts"] == r["nfe2"]
        assert np.allclose(r["saveval"][1:], acc[:, 2] * acc[:, 1]) and r["saveval"][0] == 0
        assert r["ndraws"] <= 1 + r["nattempts"]
        w, _ = o.path_total(B)
        w_all.append(w[0, 0])
        err_all.append(abs(r["u"][0, 0] - np.exp((a - 0.5 * b * b) + b * w[0, 0])))
    w_all = np.array(w_all)
    assert nrej > 20000                    # the test is about rejections: make sure there were plenty
    assert np.median(err_all) < 2e-2
    assert abs(w_all.mean()) < 4 / np.sqrt(len(w_all)) and abs(w_all.var() - 1.0) < 4 * np.sqrt(2.0 / len(w_all))


def test_shared_controller_batch_and_float32_agree():
    """Config-5 shapes at a small batch: fp32 and fp64 oracles on the same noise pool take the same accept/reject sequence
    (tol 0.14: truncation dominates rounding by orders of magnitude) and agree to fp32 rounding."""
    drift, diff = arch_nsde_drift(), arch_nsde_diffusion()
    rng = np.random.default_rng(5)
    p = nsde_params(drift, diff, rng, np.float64, 2.0, 0.5)
    x = rng.standard_normal((24, 32))
    noise = rng.standard_normal((300, 2, 24, 32))
    r64 = SdeOracle(drift, diff, np.float64, max_attempts=299).forward(x, p, noise)
    r32 = SdeOracle(drift, diff, np.float32, max_attempts=299).forward(x, p, noise)
    assert r64["rc"] == 0 and r32["rc"] == 0
    assert r64["nattempts"] == r32["nattempts"] and np.array_equal(r64["steps"][:, 3], r32["steps"][:, 3])
    assert np.abs(r32["u"] - r64["u"]).max() < 2e-4 * np.abs(r64["u"]).max()
    assert np.allclose(r32["saveval"], r64["saveval"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("name", ["SOSRI", "SRIW1"])
def test_reverse_pass_matches_finite_differences(name):
    drift = make_arch([3, 5, 3], ["tanh", "identity"], False)
    diff = make_arch([3, 3], ["identity"], False)
    rng = np.random.default_rng(7)
    B = 4
    p = nsde_params(drift, diff, rng, np.float64, 2.0, 0.6)
    p = p + 0.05 * rng.standard_normal(len(p))
    x = rng.standard_normal((B, 3))
    noise = rng.standard_normal((401, 2, B, 3))
    o = SdeOracle(drift, diff, np.float64, 0.1, 0.1, tableau=name, reg_kind=1, max_attempts=400, qmax=10.0, gamma=1.0, beta2=1e-9)
    r0 = o.forward(x, p, noise)
    assert r0["rc"] == 0 and (r0["steps"][:, 3] == 0).any(), "want a sequence with rejected steps"
    dts, acc = r0["steps"][:, 1].copy(), r0["steps"][:, 3].astype(np.int32)
    o.set_replay(dts, acc)           # freeze the sequence: step sizes are constants of the reverse pass
    r = o.forward(x, p, noise)
    assert np.array_equal(r["u"], r0["u"])
    ubar = rng.standard_normal((B, 3))
    svbar = rng.standard_normal(len(r["saveval"]))

    def loss(xx, pp):
        q = o.forward(xx, pp, noise)
        return float((q["u"] * ubar).sum() + (q["saveval"] * svbar).sum())

    o.forward(x, p, noise)
    xbar, pbar = o.backward(ubar, svbar)
    eps = 1e-6
    for idx in rng.choice(len(p), 12, replace=False):
        pp, pm = p.copy(), p.copy()
        pp[idx] += eps; pm[idx] -= eps
        fd = (loss(x, pp) - loss(x, pm)) / (2 * eps)
        assert abs(fd - pbar[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, pbar[idx])
    for idx in [(0, 0), (1, 2), (3, 1)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps; xm[idx] -= eps
        fd = (loss(xp, p) - loss(xm, p)) / (2 * eps)
        assert abs(fd - xbar[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, xbar[idx])


def test_stiffness_estimate_of_the_sri_step_and_its_reverse():
    """reg_kind = 2, the reference's shipped NSDE default (experiments/configs/mnist_nsde.yml:6, mnist_nsde.jl:51-61): the callback records
    |eigen_est| / 10.6 with eigen_est = rms(k4 - k3) / rms(H0_4 - H0_3) of the SOSRI2 step.  Pinned here by (a) a linear drift c*u, for which the
    quotient is |c| whatever the noise; (b) the definition evaluated from the attempt's own stage arrays; (c) fp64 central differences of the
    whole reverse pass along a frozen step sequence (with rejected steps in it)."""
    # (a) drift = c * u (one identity Dense with W = c I): k4 - k3 = c (H0_4 - H0_3) exactly
    c = -3.25
    drift = make_arch([3, 3], ["identity"], False)
    diff = make_arch([3, 3], ["identity"], False)
    rng = np.random.default_rng(3)
    B = 5
    p = np.concatenate([(c * np.eye(3)).reshape(-1, order="F"), np.zeros(3), 0.3 * rng.standard_normal(9), np.zeros(3)])
    x = rng.standard_normal((B, 3))
    noise = rng.standard_normal((200, 2, B, 3))
    o = SdeOracle(drift, diff, np.float64, 0.05, 0.05, tableau="SOSRI2", reg_kind=2, cb_save_start=1, max_attempts=199)
    r = o.forward(x, p, noise)
    assert r["rc"] == 0 and len(r["saveval"]) == int(r["steps"][:, 3].sum()) + 1
    assert r["saveval"][0] == 1.0 / 10.6                                  # the callback's initialisation: eigen_est = 1 (as the ODE oracle)
    assert np.allclose(r["saveval"][1:], abs(c) / 10.6, rtol=1e-10)
    nn = o.eigen_norms()
    assert nn.shape == (len(r["saveval"]) - 1, 2) and np.allclose(nn[:, 0] / nn[:, 1], abs(c), rtol=1e-10)
    o2 = SdeOracle(drift, diff, np.float64, 0.05, 0.05, tableau="SOSRI2", reg_kind=2, cb_save_start=0, max_attempts=199, stability_size=2.0)
    assert np.allclose(o2.forward(x, p, noise)["saveval"], abs(c) / 2.0, rtol=1e-10)     # the constant is a configuration value

    # (b) + (c) the experiment's shape in small: Dense(3, 5, tanh) -> Dense(5, 3), Dense(3, 3)
    drift = make_arch([3, 5, 3], ["tanh", "identity"], False)
    rng = np.random.default_rng(11)
    B = 4
    p = nsde_params(drift, diff, rng, np.float64, 2.0, 0.6)
    p = p + 0.05 * rng.standard_normal(len(p))
    x = rng.standard_normal((B, 3))
    noise = rng.standard_normal((401, 2, B, 3))
    o = SdeOracle(drift, diff, np.float64, 0.1, 0.1, tableau="SOSRI2", reg_kind=2, max_attempts=400, qmax=10.0, gamma=1.0, beta2=1e-9)
    T = sri_tableau("SOSRI2")
    dt = 0.07
    dW, dZ = np.sqrt(dt) * noise[0, 0], np.sqrt(dt) * noise[0, 1]
    kg, _, _ = o.attempt(p, x, dt, dW, dZ)
    chi2 = (dW + dZ / np.sqrt(3.0)) / 2
    H0 = [x + dt * sum(T["A0"][s, j] * kg[j] for j in range(s)) + chi2 * sum(T["B0"][s, j] * kg[4 + j] for j in range(s)) for s in range(4)]
    n1, n2 = o.eigen_norms()[0]
    assert np.isclose(n1, np.sqrt(np.mean((kg[3] - kg[2]) ** 2)), rtol=1e-12) and np.isclose(n2, np.sqrt(np.mean((H0[3] - H0[2]) ** 2)), rtol=1e-12)

    r0 = o.forward(x, p, noise)
    assert r0["rc"] == 0 and (r0["steps"][:, 3] == 0).any(), "want a sequence with rejected steps"
    o.set_replay(r0["steps"][:, 1].copy(), r0["steps"][:, 3].astype(np.int32))
    r = o.forward(x, p, noise)
    assert np.array_equal(r["u"], r0["u"]) and np.array_equal(r["saveval"], r0["saveval"]) and (r["saveval"][1:] > 0).all()
    ubar = rng.standard_normal((B, 3))
    svbar = rng.standard_normal(len(r["saveval"]))

    def loss(xx, pp):
        q = o.forward(xx, pp, noise)
        return float((q["u"] * ubar).sum() + (q["saveval"] * svbar).sum())

    o.forward(x, p, noise)
    xbar, pbar = o.backward(ubar, svbar)
    xbar0, pbar0 = o.backward(ubar, np.zeros_like(svbar))
    assert np.abs(pbar - pbar0).max() > 1e-3 * np.abs(pbar0).max()       # the term is not negligible in what is being checked
    eps = 1e-6
    for idx in rng.choice(len(p), 14, replace=False):
        pp, pm = p.copy(), p.copy()
        pp[idx] += eps; pm[idx] -= eps
        fd = (loss(x, pp) - loss(x, pm)) / (2 * eps)
        assert abs(fd - pbar[idx]) <= 2e-6 * max(1.0, abs(fd)), (idx, fd, pbar[idx])
    for idx in [(0, 0), (1, 2), (3, 1)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps; xm[idx] -= eps
        fd = (loss(xp, p) - loss(xm, p)) / (2 * eps)
        assert abs(fd - xbar[idx]) <= 2e-6 * max(1.0, abs(fd)), (idx, fd, xbar[idx])


def test_saveat_linear_interpolation_and_its_reverse():
    """The {R,true} call methods: states at the save times (linear interpolant between accepted steps, u0 at t0, the end state
    at t1), as a (B, T, D) array, and the reverse pass for a cotangent of that shape against finite differences."""
    drift = make_arch([3, 5, 3], ["tanh", "identity"], False)
    diff = make_arch([3, 3], ["identity"], False)
    rng = np.random.default_rng(17)
    B = 3
    p = nsde_params(drift, diff, rng, np.float64, 2.0, 0.6)
    x = rng.standard_normal((B, 3))
    noise = rng.standard_normal((200, 2, B, 3))
    o = SdeOracle(drift, diff, np.float64, 0.1, 0.1, reg_kind=1, max_attempts=199)
    end = o.forward(x, p, noise)
    sa = np.array([0.0, 0.13, 0.5, 0.77, 1.0])
    o.set_saveat(sa)
    r = o.forward(x, p, noise)
    assert r["u"].shape == (B, 5, 3) and r["nattempts"] == end["nattempts"]
    assert np.array_equal(r["u"][:, 0], x) and np.array_equal(r["u"][:, -1], end["u"])
    # an interior point lies on the segment between the states of the step that covers it
    acc = r["steps"][r["steps"][:, 3] == 1]
    o.set_replay(r["steps"][:, 1], r["steps"][:, 3].astype(np.int32))          # freeze the sequence for finite differences
    ubar = rng.standard_normal((B, 5, 3))
    svbar = rng.standard_normal(len(r["saveval"]))

    def loss(xx, pp):
        q = o.forward(xx, pp, noise)
        return float((q["u"] * ubar).sum() + (q["saveval"] * svbar).sum())

    o.forward(x, p, noise)
    xbar, pbar = o.backward(ubar, svbar)
    eps = 1e-6
    for idx in rng.choice(len(p), 10, replace=False):
        pp, pm = p.copy(), p.copy()
        pp[idx] += eps; pm[idx] -= eps
        fd = (loss(x, pp) - loss(x, pm)) / (2 * eps)
        assert abs(fd - pbar[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, pbar[idx])
    for idx in [(0, 0), (2, 1)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps; xm[idx] -= eps
        fd = (loss(xp, p) - loss(xm, p)) / (2 * eps)
        assert abs(fd - xbar[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, xbar[idx])


@pytest.mark.parametrize("tag,dt", [("f32", np.float32), ("f64", np.float64)])
@pytest.mark.parametrize("name", ["nsde_B8", "nsde_B5_rejecting"])
def test_sde_oracle_reproduces_golden(name, tag, dt):
    """Regression pin of the stochastic oracle on the committed fixtures (tests/golden/make_golden.py round2)."""
    import os
    from oracle.oracle_sde import SdeOracle
    from tests.golden.make_golden import nsde_inputs
    drift, diff, p, x, wu, noise, tol, ctrl = nsde_inputs(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    o = SdeOracle(drift, diff, dt, tol, tol, tableau="SOSRI", max_attempts=399, **ctrl)
    r = o.forward(x, p, noise)
    assert r["rc"] == 0 and r["nfe1"] == int(g[f"nfe1_{tag}"]) and r["ndraws"] == int(g[f"ndraws_{tag}"])
    assert np.array_equal(r["steps"][:, 3], g[f"steps_{tag}"][:, 3])
    rt = 1e-11 if dt == np.float64 else 2e-4
    np.testing.assert_allclose(r["u"], g[f"u_{tag}"], rtol=rt, atol=rt)
    xb, pb = o.backward(wu, np.full(len(r["saveval"]), 5.0))
    gt = 1e-8 if dt == np.float64 else 3e-3
    np.testing.assert_allclose(xb, g[f"xbar_{tag}"], rtol=gt, atol=gt * np.abs(g[f"xbar_{tag}"]).max())
    np.testing.assert_allclose(pb, g[f"pbar_{tag}"], rtol=gt, atol=gt * np.abs(g[f"pbar_{tag}"]).max())


@pytest.mark.parametrize("tag,dt", [("f32", np.float32), ("f64", np.float64)])
@pytest.mark.parametrize("name", ["nsde_stiff_B8", "nsde_stiff_B5_rejecting"])
def test_sde_oracle_reproduces_the_stiffness_fixtures(name, tag, dt):
    """Regression pin of the stiffness-estimate regulariser (reg_kind 2, SOSRI2) on the committed fixtures (tests/golden/make_golden.py stiff)."""
    import os
    from oracle.oracle_sde import SdeOracle
    from tests.golden.make_golden import nsde_stiff_inputs
    drift, diff, p, x, wu, noise, tol, ctrl = nsde_stiff_inputs(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    o = SdeOracle(drift, diff, dt, tol, tol, tableau="SOSRI2", reg_kind=2, max_attempts=399, **ctrl)
    r = o.forward(x, p, noise)
    assert r["rc"] == 0 and r["nfe1"] == int(g[f"nfe1_{tag}"]) and r["ndraws"] == int(g[f"ndraws_{tag}"])
    assert np.array_equal(r["steps"][:, 3], g[f"steps_{tag}"][:, 3])
    rt = 1e-11 if dt == np.float64 else 2e-4
    np.testing.assert_allclose(r["u"], g[f"u_{tag}"], rtol=rt, atol=rt)
    np.testing.assert_allclose(r["saveval"], g[f"saveval_{tag}"], rtol=10 * rt)
    nn = o.eigen_norms()
    np.testing.assert_allclose(r["saveval"][1:], (nn[:, 0] / nn[:, 1]) / 10.6, rtol=1e-6)
    xb, pb = o.backward(wu, np.full(len(r["saveval"]), 3.0))
    gt = 1e-8 if dt == np.float64 else 3e-3
    np.testing.assert_allclose(xb, g[f"xbar_{tag}"], rtol=gt, atol=gt * np.abs(g[f"xbar_{tag}"]).max())
    np.testing.assert_allclose(pb, g[f"pbar_{tag}"], rtol=gt, atol=gt * np.abs(g[f"pbar_{tag}"]).max())
