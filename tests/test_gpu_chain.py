"""GPU parity of the chain engine (csrc/rnde_chain.h: small-width Dense chains, e.g. the latent-ODE dynamics of
experiments/latent_ode.jl:113-124) against the CPU oracle, through the C ABI.

Tolerances as in test_gpu_forward.py: one f evaluation / one attempt 2e-5 absolute on O(1) values (fp32 dot products
in a different association order, tanh within 2 ulp); full solves in the truncation-dominated regime must reproduce
the oracle's accept/reject sequence exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _arch(kind):
    from tests.util import arch_mnist, arch_test_node, make_arch
    from oracle.oracle import arch_latent
    return {"latent": arch_latent,                                                            # config 4
            "chain3": lambda: make_arch([7, 33, 64, 7], ["tanh", "identity", "tanh"], True),   # ragged widths, time dependent
            "wide": lambda: make_arch([64, 64, 64], ["tanh", "tanh"], False),                  # the width limit
            "one": lambda: make_arch([5, 5], ["tanh"], True, pre_act=True),
            "test_node": arch_test_node, "small": lambda: arch_mnist(36, 10)}[kind]()


def _setup(kind, B, seed, scale=1.0):
    from tests.util import glorot_params
    rng = np.random.default_rng(seed)
    arch = _arch(kind)
    p = glorot_params(arch, rng, np.float32, scale)
    p = (p + 0.05 * rng.standard_normal(p.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, arch.dims[0])).astype(np.float32)
    return arch, p, x


def _cfg(arch, B, **kw):
    from tests.test_gpu_forward import _cfg as base
    kw.setdefault("col_tile", _DEFAULT_TILE[0])
    return base(arch, B, **kw)


KINDS = [("latent", 4), ("latent", 37), ("latent", 512), ("chain3", 19), ("wide", 16), ("one", 3), ("test_node", 5), ("small", 70)]
LAYOUTS = [64, 65]     # col_tile: one wave per 16 batch columns (rnde_chain.h) / four waves per 16 columns (rnde_chainmw.h)
_DEFAULT_TILE = [64]


@pytest.fixture(autouse=True, params=[64, 65], ids=["one-wave", "multi-wave"])
def _engine(request):
    """Every test of this file that does not pick a layout itself runs on both kernel families of the chain engine."""
    _DEFAULT_TILE[0] = request.param
    yield
    _DEFAULT_TILE[0] = 64


# tests of features that exist on the multi-wave kernels only are generated for that kernel family only (no skipped twin)
_MW = pytest.mark.parametrize("_engine", [65], indirect=True, ids=["multi-wave"])


@pytest.mark.parametrize("lay", LAYOUTS)
@pytest.mark.parametrize("kind,B", KINDS)
def test_chain_feval_matches_oracle(kind, B, lay):
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 1)
    got = Node(_cfg(arch, B, col_tile=lay)).feval(x, p, 0.37)
    assert np.abs(got - Oracle(arch, np.float64).f_eval(p, x, 0.37)).max() <= 2e-5
    assert np.abs(got - Oracle(arch, np.float32).f_eval(p, x, 0.37)).max() <= 2e-5


@pytest.mark.parametrize("lay", LAYOUTS)
@pytest.mark.parametrize("kind,B", KINDS)
def test_chain_attempt_matches_oracle(kind, B, lay):
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 2)
    o64 = Oracle(arch, np.float64, reltol=1e-6, abstol=1e-6)
    k1 = o64.f_eval(p, x, 0.1).astype(np.float32)
    t, dt = 0.1, 0.05
    kref, unew_ref, eest_ref, _ = o64.attempt(p, x, k1, t, dt)
    kout, unew, eest = Node(_cfg(arch, B, reltol=1e-6, abstol=1e-6, col_tile=lay)).attempt(x, k1, p, t, dt)
    assert np.abs(kout - kref).max() <= 2e-5
    assert np.abs(unew - unew_ref).max() <= 2e-5
    floor = 3 * 6e-8 * dt * np.abs(kref).max() / 1e-6      # fp32 rounding floor of EEst (see test_gpu_forward.py)
    assert abs(eest - eest_ref) <= 5e-3 * eest_ref + floor


@pytest.mark.parametrize("kind,B,tol,scale", [("latent", 4, 1e-3, 2.0), ("latent", 100, 1e-4, 2.0), ("chain3", 19, 1e-3, 2.0),
                                               ("wide", 16, 1e-3, 1.5), ("one", 3, 1e-3, 3.0), ("test_node", 5, 1e-3, 3.0),
                                               ("small", 70, 1e-3, 4.0)])
@pytest.mark.parametrize("lay", LAYOUTS)
def test_chain_solve_matches_oracle(kind, B, tol, scale, lay):
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 3, scale)
    ref = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1).forward(x, p)
    got = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=lay)).forward(x, p)
    assert got["nfe"] == ref["nfe"] and (got["steps"][:, 3] == ref["steps"][:, 3]).all()
    # conditioning of each column: 8 tanh layers with scaled weights amplify fp32 rounding along some trajectories by 1e4;
    # the spread between the fp32 and the fp64 oracle measures that, and the device is held to the same spread
    r32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1).forward(x, p)
    spread = np.abs(r32["u"] - ref["u"]).max(axis=1) if r32["nfe"] == ref["nfe"] else np.full(B, 1e-2)
    err = np.abs(got["u"] - ref["u"]).max(axis=1)
    assert (err <= 3e-5 * max(1.0, np.abs(ref["u"]).max()) + 4 * spread).all()
    np.testing.assert_allclose(got["saveval"], ref["saveval"], rtol=5e-2, atol=1e-6)


@pytest.mark.parametrize("kind,B,tol,scale,saveat", [("latent", 4, 1e-3, 2.0, np.linspace(0, 1, 49)), ("latent", 70, 1e-4, 2.0, np.array([0.1, 0.5, 0.9])),
                                                      ("chain3", 19, 1e-3, 2.0, np.array([0.0, 0.25, 1.0]))])
@pytest.mark.parametrize("lay", LAYOUTS)
def test_chain_saveat_matches_oracle(kind, B, tol, scale, saveat, lay):
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 5, scale)
    sa = saveat.astype(np.float32)
    ref = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1).forward(x, p, saveat=sa)
    got = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=lay)).forward_saveat(x, p, sa)
    assert got["nfe"] == ref["nfe"]
    assert got["u"].shape == ref["u"].shape == (B, len(sa), arch.dims[0])
    r32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1).forward(x, p, saveat=sa)
    spread = np.abs(r32["u"] - ref["u"]).max(axis=(1, 2)) if r32["nfe"] == ref["nfe"] else np.full(B, 1e-2)
    err = np.abs(got["u"] - ref["u"]).max(axis=(1, 2))
    assert (err <= 3e-5 * max(1.0, np.abs(ref["u"]).max()) + 8 * spread).all()


def test_chain_config4_statistics():
    """SURVEY.md 8d config 4 at full size (D = 20, 8 layers, B = 512, 49 save points, tol 1.4e-8): at this tolerance fp32 EEst
    sits on its rounding floor, so the attempt count is compared statistically and the states against the
    fp64 oracle."""
    from tests.util import Node, Oracle
    arch, p, x = _setup("latent", 512, 7, 1.0)
    sa = np.linspace(0, 1, 49).astype(np.float32)
    ref64 = Oracle(arch, np.float64, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1).forward(x, p, saveat=sa)
    ref32 = Oracle(arch, np.float32, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1).forward(x, p, saveat=sa)
    got = Node(_cfg(arch, 512, max_attempts=512)).forward_saveat(x, p, sa)
    n_dev, n32, n64 = (got["nfe"] - 3) // 6, ref32["nattempts"], ref64["nattempts"]
    print(f"attempts: device {n_dev}, oracle f32 {n32}, oracle f64 {n64}")
    assert abs(n_dev - n32) <= 0.25 * n32 + 1
    assert np.abs(got["u"] - ref64["u"]).max() <= 1e-5 * max(1.0, np.abs(ref64["u"]).max())
    # round 3: the oracle in the DEVICE'S summation order (Oracle(sum_order=3): per output two interleaved accumulators over the 16-wide
    # k-blocks, acc0 seeded with the bias, K = 4 FMA chains; the device's tanh formula) -- the natural run's attempt count is then an
    # equality (+-1), as for the MNIST shape (tests/test_gpu_replay.py)
    o3 = Oracle(arch, np.float32, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1, sum_order=3, max_attempts=512)
    r3 = o3.forward(x, p, saveat=sa)
    n3 = r3["nattempts"]
    print(f"attempts: device {n_dev}, device-order oracle {n3}; u vs that oracle {np.abs(got['u'] - r3['u']).max():.2e}")
    assert abs(n_dev - n3) <= 1
    assert np.abs(got["u"] - r3["u"]).max() <= 3e-6 * max(1.0, np.abs(r3["u"]).max())


@_MW
@pytest.mark.parametrize("kind,B", [("latent", 64), ("chain3", 19), ("wide", 16)])
def test_chain_feval_equals_the_device_order_oracle_almost_bit_for_bit(kind, B, _mw_only):
    """One f evaluation on the multi-wave chain kernels against the oracle in their order: bit-identical in most entries, 1-2 ulp in the
    rest (v_exp_f32 / v_rcp_f32 against correctly rounded exp2 / reciprocal, once per tanh layer)."""
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 4, 1.0)
    fd = Node(_cfg(arch, B)).feval(x, p, 0.37)
    f0 = Oracle(arch, np.float32, sum_order=0).f_eval(p, x, 0.37)
    f3 = Oracle(arch, np.float32, sum_order=3).f_eval(p, x, 0.37)
    eq0, eq3 = float(np.mean(fd == f0)), float(np.mean(fd == f3))
    print(f"{kind}: entries bit-equal to the device: sequential-k oracle {eq0:.3f}, device-order oracle {eq3:.3f}; max |diff| {np.abs(fd - f0).max():.2e} / {np.abs(fd - f3).max():.2e}")
    assert eq3 >= 0.6 and eq3 > eq0 and np.abs(fd - f3).max() <= 4e-7 * max(1.0, np.abs(f3).max())


@pytest.mark.parametrize("kind,B,tol,scale,ws", [("latent", 4, 1e-3, 1.5, 0.0), ("latent", 37, 1e-3, 1.5, 30.0), ("chain3", 19, 1e-3, 2.0, 30.0),
                                                  ("wide", 16, 1e-3, 1.5, 30.0), ("one", 3, 1e-3, 3.0, 30.0), ("test_node", 5, 1e-3, 3.0, 30.0),
                                                  ("small", 70, 1e-3, 4.0, 30.0)])
def test_chain_reverse_matches_oracle(kind, B, tol, scale, ws):
    """x-bar, p-bar, tspan-bar of the chain engine against the fp64 oracle (same accept/reject sequence); tolerance 2e-3 of the
    largest entry plus the fp32-vs-fp64 oracle spread (conditioning of the case)."""
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 3, scale)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1)
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1)
    r64, r32 = o64.forward(x, p), o32.forward(x, p)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol))
    got = node.forward(x, p, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"]
    rng = np.random.default_rng(11)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    svbar = np.full(len(got["saveval"]), ws, dtype=np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, t32 = o32.backward(ubar, svbar)
    cx, cp = rel_err(x32, x64), rel_err(p32, p64)
    print(f"{kind}: x-bar {rel_err(gx, x64):.2e} (oracle spread {cx:.2e})  p-bar {rel_err(gp, p64):.2e} ({cp:.2e})  tspan {gt} vs {t64}")
    assert rel_err(gx, x64) <= 2e-3 + 4 * cx
    assert rel_err(gp, p64) <= 2e-3 + 4 * cp
    assert np.abs(gt - t64).max() <= (2e-3 + 4 * max(cx, cp)) * max(1.0, np.abs(t64).max())


@pytest.mark.parametrize("kind,B,tol,scale,saveat", [("latent", 4, 1e-3, 1.5, np.linspace(0, 1, 49)), ("latent", 70, 1e-4, 1.5, np.array([0.1, 0.5, 0.9])),
                                                      ("chain3", 19, 1e-3, 2.0, np.array([0.0, 0.25, 1.0]))])
def test_chain_saveat_reverse_matches_oracle(kind, B, tol, scale, saveat):
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 5, scale)
    sa = saveat.astype(np.float32)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1)
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1)
    r64, r32 = o64.forward(x, p, saveat=sa), o32.forward(x, p, saveat=sa)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol))
    got = node.forward_saveat(x, p, sa, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"]
    rng = np.random.default_rng(12)
    ubar = rng.standard_normal(r64["u"].shape).astype(np.float32)
    svbar = (10 * rng.standard_normal(len(got["saveval"]))).astype(np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, t32 = o32.backward(ubar, svbar)
    cx, cp = rel_err(x32, x64), rel_err(p32, p64)
    print(f"{kind}: x-bar {rel_err(gx, x64):.2e} ({cx:.2e})  p-bar {rel_err(gp, p64):.2e} ({cp:.2e})  tspan {gt} vs {t64}")
    assert rel_err(gx, x64) <= 2e-3 + 4 * cx
    assert rel_err(gp, p64) <= 2e-3 + 4 * cp
    assert np.abs(gt - t64).max() <= (2e-3 + 4 * max(cx, cp)) * max(1.0, np.abs(t64).max())


def test_chain_config4_reverse():
    """Config 4 at full size, tol 1.4e-8: both solves are converged to ~1e-7, so the cotangents of the saved states agree with
    the fp64 oracle's even though the step sequences differ (fp32 noise floor); 1e-3 of the largest entry."""
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup("latent", 512, 7, 1.0)
    sa = np.linspace(0, 1, 49).astype(np.float32)
    o64 = Oracle(arch, np.float64, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1)
    r64 = o64.forward(x, p, saveat=sa)
    node = Node(_cfg(arch, 512, max_attempts=512))
    got = node.forward_saveat(x, p, sa, keep_tape=True)
    rng = np.random.default_rng(13)
    ubar = rng.standard_normal(r64["u"].shape).astype(np.float32)
    gx, gp, gt = node.backward(ubar, None)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), None)
    print(f"x-bar {rel_err(gx, x64):.2e}  p-bar {rel_err(gp, p64):.2e}  tspan {gt} vs {t64}")
    assert rel_err(gx, x64) <= 1e-3 and rel_err(gp, p64) <= 1e-3


def test_chain_generic_dispatch_path_on_latent_shape(monkeypatch):
    """The latent-ODE widths normally take the compile-time-shape kernels (rnde_chain.h: ALT); RNDE_CHAIN_GENERIC=1 forces the
    run-time-dispatch kernels on the same network: both must agree with the oracle (forward and reverse)."""
    from tests.util import Node, Oracle, rel_err
    monkeypatch.setenv("RNDE_CHAIN_GENERIC", "1")
    arch, p, x = _setup("latent", 37, 3, 1.5)
    sa = np.array([0.0, 0.3, 0.8, 1.0], dtype=np.float32)
    o64 = Oracle(arch, np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    o32 = Oracle(arch, np.float32, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r64, r32 = o64.forward(x, p, saveat=sa), o32.forward(x, p, saveat=sa)
    node = Node(_cfg(arch, 37, reltol=1e-3, abstol=1e-3))
    got = node.forward_saveat(x, p, sa, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"]
    spread = np.abs(r32["u"] - r64["u"]).max(axis=(1, 2))
    assert (np.abs(got["u"] - r64["u"]).max(axis=(1, 2)) <= 3e-5 * max(1.0, np.abs(r64["u"]).max()) + 4 * spread).all()
    rng = np.random.default_rng(14)
    ubar = rng.standard_normal(r64["u"].shape).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 20.0, dtype=np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, _ = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, _ = o32.backward(ubar, svbar)
    assert rel_err(gx, x64) <= 2e-3 + 4 * rel_err(x32, x64)
    assert rel_err(gp, p64) <= 2e-3 + 4 * rel_err(p32, p64)


@pytest.mark.parametrize("kind,B,tol,scale", [("latent", 21, 1e-3, 1.5), ("chain3", 19, 1e-3, 2.0), ("small", 12, 1e-3, 4.0)])
@pytest.mark.parametrize("reg,agg", [(2, "max"), (2, "mean"), (3, "mean"), (4, "mean")])      # 4: |eigen_est * dt|, the callback of test/test_node.jl:75
def test_chain_stiffness_regulariser_matches_oracle(kind, B, tol, scale, reg, agg):
    """regularize = stiff_est / error_stiff_est on the chain engine (latent_ode.jl:127-136 selects AutoTsit5(Tsit5()) for them):
    callback values and the reverse pass of eigen_est = ||k7-k6|| / ||u-g6|| against the oracle."""
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 4, scale)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, regularize=reg))
    got = node.forward(x, p, keep_tape=True)
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=reg)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=reg)
    r32, r64 = o32.forward(x, p), o64.forward(x, p)
    assert got["nfe"] == r32["nfe"] == r64["nfe"]
    np.testing.assert_allclose(got["saveval"], r64["saveval"], rtol=5e-2, atol=1e-5)
    rng = np.random.default_rng(7)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    svbar = np.zeros(len(got["saveval"]), dtype=np.float32)
    if agg == "max":
        svbar[int(np.argmax(r64["saveval"]))] = 2.0
    else:
        svbar[:] = 2.0 / len(svbar)
    xb, pb, tsb = node.backward(ubar, svbar)
    xb64, pb64, _ = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    xb32, pb32, _ = o32.backward(ubar, svbar)
    cx, cp = rel_err(xb32, xb64), rel_err(pb32, pb64)
    print(f"reg {reg}/{agg} {kind}: x-bar {rel_err(xb, xb64):.2e} (oracle f32 {cx:.2e})  p-bar {rel_err(pb, pb64):.2e} (oracle f32 {cp:.2e})")
    assert rel_err(xb, xb64) <= 3e-3 + 4 * cx
    assert rel_err(pb, pb64) <= 3e-3 + 4 * cp


# ---- explicit RK pair as data (rnde_chainmw.h RkTab): Dormand-Prince 5(4) on the device against the DP5 oracle (itself pinned
# ---- step for step on scipy's RK45, tests/test_oracle.py), and Tsit5 through the same data path against the constant-folded kernels
DP5_CASES = [("latent", 4, 1e-3, 2.0), ("latent", 100, 1e-4, 2.0), ("chain3", 19, 1e-3, 2.0), ("wide", 16, 1e-3, 1.5), ("one", 3, 1e-3, 3.0),
             ("test_node", 5, 1e-3, 3.0), ("small", 70, 1e-3, 4.0)]


@pytest.fixture
def _mw_only():
    assert _DEFAULT_TILE[0] == 65


@_MW
@pytest.mark.parametrize("kind,B", [("latent", 37), ("chain3", 19), ("small", 70)])
def test_dp5_attempt_matches_oracle(kind, B, _mw_only):
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 2)
    o64 = Oracle(arch, np.float64, reltol=1e-6, abstol=1e-6, solver="DP5")
    k1 = o64.f_eval(p, x, 0.1).astype(np.float32)
    t, dt = 0.1, 0.05
    kref, unew_ref, eest_ref, _ = o64.attempt(p, x, k1, t, dt)
    kout, unew, eest = Node(_cfg(arch, B, reltol=1e-6, abstol=1e-6, solver="DP5")).attempt(x, k1, p, t, dt)
    assert np.abs(kout - kref).max() <= 2e-5
    assert np.abs(unew - unew_ref).max() <= 2e-5
    floor = 3 * 6e-8 * dt * np.abs(kref).max() / 1e-6
    assert abs(eest - eest_ref) <= 5e-3 * eest_ref + floor
    # and it is a different method from Tsit5 (guards against the table being ignored)
    _, unew_ts, eest_ts, _ = Oracle(arch, np.float64, reltol=1e-6, abstol=1e-6).attempt(p, x, k1, t, dt)
    assert abs(eest_ts - eest_ref) > 0.05 * eest_ref


@_MW
@pytest.mark.parametrize("kind,B,tol,scale", DP5_CASES)
def test_dp5_solve_and_reverse_match_oracle(kind, B, tol, scale, _mw_only):
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 3, scale)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1, solver="DP5")
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, solver="DP5")
    r64, r32 = o64.forward(x, p), o32.forward(x, p)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, solver="DP5"))
    got = node.forward(x, p, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"] and (got["steps"][:, 3] == r64["steps"][:, 3]).all()
    spread = np.abs(r32["u"] - r64["u"]).max(axis=1)
    assert (np.abs(got["u"] - r64["u"]).max(axis=1) <= 3e-5 * max(1.0, np.abs(r64["u"]).max()) + 4 * spread).all()
    np.testing.assert_allclose(got["saveval"], r64["saveval"], rtol=5e-2, atol=1e-6)
    rng = np.random.default_rng(11)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 30.0, dtype=np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, _ = o32.backward(ubar, svbar)
    cx, cp = rel_err(x32, x64), rel_err(p32, p64)
    print(f"DP5 {kind}: x-bar {rel_err(gx, x64):.2e} ({cx:.2e})  p-bar {rel_err(gp, p64):.2e} ({cp:.2e})  tspan {gt} vs {t64}")
    assert rel_err(gx, x64) <= 2e-3 + 4 * cx
    assert rel_err(gp, p64) <= 2e-3 + 4 * cp
    assert np.abs(gt - t64).max() <= (2e-3 + 4 * max(cx, cp)) * max(1.0, np.abs(t64).max())


@_MW
@pytest.mark.parametrize("kind,B,tol,scale,saveat", [("latent", 4, 1e-3, 1.5, np.linspace(0, 1, 49)), ("latent", 70, 1e-4, 1.5, np.array([0.1, 0.5, 0.9])),
                                                      ("chain3", 19, 1e-3, 2.0, np.array([0.0, 0.25, 1.0]))])
def test_dp5_saveat_and_reverse_match_oracle(kind, B, tol, scale, saveat, _mw_only):
    """Dense output of the pair (Shampine's quartic for DP5) forward and through the reverse pass."""
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 5, scale)
    sa = saveat.astype(np.float32)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1, solver="DP5")
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, solver="DP5")
    r64, r32 = o64.forward(x, p, saveat=sa), o32.forward(x, p, saveat=sa)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, solver="DP5"))
    got = node.forward_saveat(x, p, sa, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"]
    spread = np.abs(r32["u"] - r64["u"]).max(axis=(1, 2))
    assert (np.abs(got["u"] - r64["u"]).max(axis=(1, 2)) <= 3e-5 * max(1.0, np.abs(r64["u"]).max()) + 8 * spread).all()
    rng = np.random.default_rng(12)
    ubar = rng.standard_normal(r64["u"].shape).astype(np.float32)
    svbar = (10 * rng.standard_normal(len(got["saveval"]))).astype(np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, _ = o32.backward(ubar, svbar)
    cx, cp = rel_err(x32, x64), rel_err(p32, p64)
    print(f"DP5 saveat {kind}: x-bar {rel_err(gx, x64):.2e} ({cx:.2e})  p-bar {rel_err(gp, p64):.2e} ({cp:.2e})")
    assert rel_err(gx, x64) <= 2e-3 + 4 * cx
    assert rel_err(gp, p64) <= 2e-3 + 4 * cp
    assert np.abs(gt - t64).max() <= (2e-3 + 4 * max(cx, cp)) * max(1.0, np.abs(t64).max())


@_MW
def test_tsit5_through_the_table_equals_the_constant_folded_kernels(monkeypatch, _mw_only):
    """RNDE_CHAIN_TAB=1 feeds the Tsit5 coefficients to the data path (dense output expanded to monomials): same accept/reject
    sequence, states and cotangents as the kernels with the pair folded into the code."""
    from tests.util import Node
    arch, p, x = _setup("latent", 70, 5, 1.5)
    sa = np.linspace(0, 1, 9).astype(np.float32)
    rng = np.random.default_rng(15)

    def run():
        node = Node(_cfg(arch, 70, reltol=1e-4, abstol=1e-4))
        got = node.forward_saveat(x, p, sa, keep_tape=True)
        ubar = np.random.default_rng(16).standard_normal(got["u"].shape).astype(np.float32)
        svbar = np.full(len(got["saveval"]), 5.0, dtype=np.float32)
        return got, node.backward(ubar, svbar)

    a, ga = run()
    monkeypatch.setenv("RNDE_CHAIN_TAB", "1")
    b, gb = run()
    assert a["nfe"] == b["nfe"]
    np.testing.assert_allclose(a["saveval"], b["saveval"], rtol=1e-3, atol=1e-7)
    assert np.abs(a["u"] - b["u"]).max() <= 2e-6 * max(1.0, np.abs(a["u"]).max())
    for u, v in zip(ga, gb):
        assert np.abs(u - v).max() <= 2e-5 * max(1.0, np.abs(u).max())


def test_dp5_is_refused_where_the_table_kernels_do_not_run():
    from tests.util import Node, arch_mnist
    with pytest.raises(Exception):
        Node(_cfg(_arch("latent"), 8, col_tile=64, solver="DP5"))          # one-wave kernels fold Tsit5 into the code
    with pytest.raises(Exception):
        Node(_cfg(_arch("latent"), 8, regularize=2, solver="DP5"))         # stiffness callbacks are wired for Tsit5 only
    with pytest.raises(Exception):
        Node(_cfg(arch_mnist(), 8, col_tile=16, solver="DP5"))             # stage engine


@_MW
def test_latent_shape_kernels_equal_the_generic_multi_wave_kernels(monkeypatch, _mw_only):
    """The latent-ODE shape runs kernels with register-stationary weights (rnde_chainmw.h: LAT); RNDE_CHAIN_LAT=0 keeps the generic
    multi-wave kernels on the same network: same arithmetic in the same order -> the same bits, forward and reverse."""
    from tests.util import Node
    arch, p, x = _setup("latent", 70, 5, 1.5)
    sa = np.linspace(0, 1, 9).astype(np.float32)

    def run():
        node = Node(_cfg(arch, 70, reltol=1e-4, abstol=1e-4))
        got = node.forward_saveat(x, p, sa, keep_tape=True)
        ubar = np.random.default_rng(16).standard_normal(got["u"].shape).astype(np.float32)
        return got, node.backward(ubar, np.full(len(got["saveval"]), 5.0, dtype=np.float32))

    a, ga = run()
    monkeypatch.setenv("RNDE_CHAIN_LAT", "0")
    b, gb = run()
    assert a["nfe"] == b["nfe"] and np.array_equal(a["u"], b["u"]) and np.array_equal(a["saveval"], b["saveval"])
    for u, v in zip(ga, gb):
        assert np.array_equal(u, v)


# ---- an S-stage pair as a table (round 3): RNDE_SOLVER_DOP853 = scipy's DOP853 coefficients as a 13-stage first-same-as-last pair ----
@_MW
@pytest.mark.parametrize("kind,B", [("latent", 37), ("chain3", 19), ("small", 70)])
def test_dop853_attempt_matches_oracle(kind, B, _mw_only):
    """One attempted step of the 13-stage table on the device against the fp64 oracle (which reproduces scipy's rk_step to 1e-14,
    tests/test_oracle.py): all twelve new stage values, u_new, the linear fifth-order error estimate."""
    from tests.util import Node, Oracle
    arch, p, x = _setup(kind, B, 2)
    o64 = Oracle(arch, np.float64, reltol=1e-6, abstol=1e-6, solver="DOP853")
    k1 = o64.f_eval(p, x, 0.1).astype(np.float32)
    t, dt = 0.1, 0.2
    kref, unew_ref, eest_ref, _ = o64.attempt(p, x, k1, t, dt)
    kout, unew, eest = Node(_cfg(arch, B, reltol=1e-6, abstol=1e-6, solver="DOP853")).attempt(x, k1, p, t, dt)
    assert kout.shape[0] == 12
    assert np.abs(kout - kref).max() <= 3e-5
    assert np.abs(unew - unew_ref).max() <= 3e-5
    floor = 3 * 6e-8 * dt * np.abs(kref).max() / 1e-6
    assert abs(eest - eest_ref) <= 2e-2 * eest_ref + floor
    _, _, eest_ts, _ = Oracle(arch, np.float64, reltol=1e-6, abstol=1e-6).attempt(p, x, k1, t, dt)
    assert abs(eest_ts - eest_ref) > 0.05 * eest_ref          # (a different method from Tsit5: the table is not ignored)


@_MW
@pytest.mark.parametrize("kind,B,tol,scale", [("latent", 70, 1e-4, 1.5), ("chain3", 33, 1e-4, 2.0), ("small", 20, 1e-5, 3.0), ("latent", 512, 1e-4, 1.5)])
def test_dop853_solve_and_reverse_match_oracle(kind, B, tol, scale, _mw_only):
    """The adaptive solve (controller exponents of order 8 from the table, NFE = 3 + 12 per attempt) and its reverse pass -- stage loops,
    tape records and evaluation counts all run on the table's stage count -- against the fp64 oracle."""
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, 3, scale)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1, solver="DOP853")
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1, solver="DOP853")
    r64, r32 = o64.forward(x, p), o32.forward(x, p)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, solver="DOP853"))
    got = node.forward(x, p, keep_tape=True)
    assert got["nfe"] == 3 + 12 * got["nattempts"]
    assert got["nfe"] == r64["nfe"] == r32["nfe"] and (got["steps"][:, 3] == r64["steps"][:, 3]).all()
    spread = np.abs(r32["u"] - r64["u"]).max(axis=1)
    assert (np.abs(got["u"] - r64["u"]).max(axis=1) <= 3e-5 * max(1.0, np.abs(r64["u"]).max()) + 4 * spread).all()
    # EEst * dt per accepted step: the estimate is a cancellation of twelve O(1) terms, in fp32 good to ~1e-5 absolute at these step sizes
    np.testing.assert_allclose(got["saveval"], r64["saveval"], rtol=8e-2, atol=5e-3 * float(r64["saveval"].max()))
    rng = np.random.default_rng(11)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 30.0, dtype=np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, t64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, _ = o32.backward(ubar, svbar)
    cx, cp = rel_err(x32, x64), rel_err(p32, p64)
    print(f"DOP853 {kind}: attempts {got['nattempts']}  x-bar {rel_err(gx, x64):.2e} ({cx:.2e})  p-bar {rel_err(gp, p64):.2e} ({cp:.2e})  tspan {gt} vs {t64}")
    assert rel_err(gx, x64) <= 2e-3 + 4 * cx
    assert rel_err(gp, p64) <= 2e-3 + 4 * cp
    # (the time cotangents collect the error estimate's noise of all thirteen stages: the one-launch solve and the launch-per-attempt path,
    #  two compilations of the same arithmetic, differ by 1.5e-2 of the largest entry here -- either side of the fp64 value)
    assert np.abs(gt - t64).max() <= (5e-3 + 8 * max(cx, cp)) * max(1.0, np.abs(t64).max())


def test_dop853_is_refused_where_the_table_has_nothing_to_offer():
    from tests.util import Node, arch_mnist
    arch, p, x = _setup("latent", 8, 1)
    assert Node(_cfg(arch, 8, col_tile=65, solver="DOP853")).forward(x, p)["nfe"] % 12 == 3                           # (the plain call works)
    with pytest.raises(Exception):
        Node(_cfg(arch, 8, col_tile=65, solver="DOP853")).forward_saveat(x, p, np.array([0.5, 1.0], np.float32))      # no dense output in the table
    with pytest.raises(Exception):
        Node(_cfg(arch, 8, col_tile=65, regularize=2, solver="DOP853"))                                  # no stiffness estimate
    with pytest.raises(Exception):
        Node(_cfg(arch, 8, col_tile=64, solver="DOP853"))                                                # one-wave kernels fold Tsit5 in
    with pytest.raises(Exception):
        Node(_cfg(arch_mnist(), 8, col_tile=16, solver="DOP853"))                                        # stage engine


@_MW
@pytest.mark.parametrize("kind,B,tol,scale,saveat,reg", [("latent", 512, 1.4e-8, 1.0, np.linspace(0, 1, 49), 1), ("chain3", 19, 1e-3, 2.0, None, 1),
                                                          ("test_node", 5, 1e-2, 8.0, np.array([0.5, 1.0]), 1), ("latent", 70, 1e-4, 1.5, None, 3),
                                                          # more than 32 column tiles (round 4): the workgroups spread over the chip and meet through agent-scope entries
                                                          ("latent", 1000, 1e-5, 1.0, np.linspace(0, 1, 7), 1), ("latent", 4096, 1.4e-8, 1.0, np.linspace(0, 1, 49), 1),
                                                          ("chain3", 2049, 1e-3, 2.0, None, 3)])
def test_one_launch_solve_is_bit_identical_to_one_launch_per_attempt(kind, B, tol, scale, saveat, reg, monkeypatch, _mw_only):
    """rnde_chainmw_kernel<.., MW_SOLVE>: the whole adaptive solve of the chain engine in ONE launch (attempt loop, controller, saveat
    bookkeeping inside the kernel; the <= 32 workgroups pinned to one XCD meet once per attempt through its L2, round 3) against the
    one-launch-per-attempt path (RNDE_CHAIN_SOLVE=0): the norm sums are formed in the same order, so step log, states, saved values and --
    through the tape -- the reverse pass must agree bit for bit, rejected steps and the reference tolerance included."""
    from tests.util import Node
    arch, p, x = _setup(kind, B, 5, scale)
    outs = []
    for one in ("1", "0"):
        monkeypatch.setenv("RNDE_CHAIN_SOLVE", one)
        node = Node(_cfg(arch, B, reltol=tol, abstol=tol, max_attempts=256, regularize=reg))
        got = node.forward(x, p, keep_tape=True) if saveat is None else node.forward_saveat(x, p, saveat.astype(np.float32), keep_tape=True)
        ubar = np.random.default_rng(9).standard_normal(got["u"].shape).astype(np.float32)
        g = node.backward(ubar, np.full(len(got["saveval"]), 3.0, dtype=np.float32))
        assert node.L.rnde_node_fallback_count(node.h) == 0
        plain = node.forward(x, p)                      # an untaped solve on the same handle afterwards
        outs.append((got, g, plain))
        node.close()
    (a, ga, pa), (b, gb, pb) = outs
    assert a["nfe"] == b["nfe"] and a["nfe"] > 9 and np.array_equal(a["u"], b["u"]) and np.array_equal(a["saveval"], b["saveval"])
    assert np.array_equal(pa["steps"], pb["steps"]) and np.array_equal(pa["u"], pb["u"])
    for u, v in zip(ga, gb):
        assert np.array_equal(u, v)


@_MW
@pytest.mark.parametrize("kind,B,tol,scale,saveat,reg", [("latent", 512, 1.4e-8, 1.0, np.linspace(0, 1, 49), 1), ("chain3", 19, 1e-3, 2.0, None, 1),
                                                          ("test_node", 5, 1e-2, 8.0, np.array([0.5, 1.0]), 1), ("latent", 70, 1e-4, 1.5, None, 3),
                                                          ("wide", 40, 1e-5, 1.5, None, 0),
                                                          ("latent", 1000, 1e-5, 1.0, np.linspace(0, 1, 7), 1), ("latent", 4096, 1.4e-8, 1.0, np.linspace(0, 1, 49), 1)])
def test_one_launch_reverse_sweep_is_bit_identical_to_one_launch_per_attempt(kind, B, tol, scale, saveat, reg, monkeypatch, _mw_only):
    """rnde_bchainmw_kernel<.., SWEEP>: the reverse sweep of the chain engine in ONE launch (the loop over the attempted steps inside the kernel,
    weights loaded once, the scalar chain carried in registers, the three partial sums of an attempt exchanged through the XCD's L2) against
    the one-launch-per-reversed-attempt path (RNDE_CHAIN_BSWEEP=0): the same sums in the same order, so every gradient must agree bit for bit
    -- rejected steps, saveat cotangents, the stiffness-estimate regulariser and the reference tolerance included."""
    from tests.util import Node
    arch, p, x = _setup(kind, B, 5, scale)
    outs = []
    for one in ("1", "0"):
        monkeypatch.setenv("RNDE_CHAIN_BSWEEP", one)
        node = Node(_cfg(arch, B, reltol=tol, abstol=tol, max_attempts=256, regularize=reg))
        got = node.forward(x, p, keep_tape=True) if saveat is None else node.forward_saveat(x, p, saveat.astype(np.float32), keep_tape=True)
        ubar = np.random.default_rng(9).standard_normal(got["u"].shape).astype(np.float32)
        g = node.backward(ubar, np.full(len(got["saveval"]), 3.0, dtype=np.float32))
        assert node.L.rnde_node_fallback_count(node.h) == 0
        outs.append((got, g))
        node.close()
    (a, ga), (b, gb) = outs
    assert a["nfe"] == b["nfe"] and a["nfe"] > 9
    print("nfe", a["nfe"])
    for u, v in zip(ga, gb):
        assert np.array_equal(u, v)
