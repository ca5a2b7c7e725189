"""Parity at the HEADLINE configuration (MNIST shape, B = 512, reltol = abstol = 1.4e-8; reference
experiments/mnist_node.jl:113-124) through the C ABI, forward AND reverse.

At that tolerance the fp32 error estimate is rounding noise (fp32 oracle: EEst 0.04-0.14 per attempt, fp64 oracle along
the same steps: 4e-8-3e-5), so every fp32 implementation picks its own step sequence and two natural runs cannot be
compared attempt by attempt.  These tests therefore REPLAY the fp32 oracle's own sequence of (proposed dt, accept) on the
device (rnde_node_forward_replay) and on the fp64 oracle (orc_set_replay), which makes everything that does not depend
on the noise comparable element-wise:

  * trajectory: u_end device vs fp64 oracle <= 2e-6 of max|u| (observed ~5e-7, the same as fp32 oracle vs fp64);
  * "exact" gradient path (cotangent on u_end only, controller and initial-step tracking off): x_bar, p_bar vs the fp64
    oracle <= 2e-5 of the largest entry (observed ~1e-6);
  * noise-defined quantities -- per-attempt EEst, saveval = EEst*dt and the gradient THROUGH them (regulariser cotangent,
    controller chain) -- are compared with the fp32 oracle statistically: the per-attempt EEst ratio is constant to 10 %
    (the device's split-K sums are more accurate than the oracle's sequential ones: its floor is ~0.46 of the oracle's), and the
    device's distance to the fp64 gradient bounded by a multiple of the fp32 oracle's own distance to it (the case's fp32
    spread).  Stated, asserted and printed; DESIGN.md section 3 lists the measured values.

Also here: the vanilla (regularize = 0) B = 64 solve of BASELINE config 1 against the oracle, and the distribution of the
number of attempts over 16 seeds, device vs fp32 oracle.

Round 3 -- the oracle in the DEVICE'S ORDER (`Oracle(sum_order=3)`, oracle/rnde_oracle.c `orc_set_sum_order`): the two Dense
layers accumulated exactly as the stage engine does (7 split-K row-block partials of two interleaved K = 4 FMA chains -- an fp32
MFMA is four fused multiply-adds in k order, tools/mfma_model.py fits that to raw matrix-pipe output bit for bit -- added in order
r = 0..6; layer 2 likewise over [h; t; 1]) and tanh by the device's formula.  With that the fp32 noise floor is the SAME noise:
one f evaluation agrees bit for bit in ~95 % of its entries (1 ulp in the rest: v_exp_f32 / v_rcp_f32 against correctly rounded
exp2 / reciprocal), natural runs take the SAME number of attempts (B = 64 x 16 seeds and B = 512: 30 and 30; the sequential-k
oracle: 40.8), per-attempt EEst agrees to 0.6 % at B = 512 (10 % at B = 64 per attempt, 0.5 % in the mean).  So the -27 % NFE
offset of round 2 IS the summation order, and natural-run NFE parity is now an equality test.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B512, TOL = 512, 1.4e-8


def _problem(B, seed):
    from tests.util import arch_mnist, glorot_params
    rng = np.random.default_rng(seed)
    arch = arch_mnist()
    p = glorot_params(arch, rng, np.float32, 1.0)
    x = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ubar = (rng.standard_normal((B, 784)) / B).astype(np.float32)
    return arch, p, x, ubar


def _cfg(B, **kw):
    from tests.util import make_cfg
    return make_cfg([784, 100, 784], ["tanh", "tanh"], B, reltol=TOL, abstol=TOL, max_attempts=96, **kw)


_cache = {}


def _oracle_runs(B, seed, track, sum_order=0):
    """fp32 oracle natural run + its reverse; fp64 oracle replayed along the same sequence + its reverse.
    sum_order = 3: the fp32 oracle accumulates and rounds as the device does (module docstring)."""
    key = (B, seed, track, sum_order)
    if key in _cache:
        return _cache[key]
    from tests.util import Oracle
    arch, p, x, ubar = _problem(B, seed)
    o32 = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, track_ctrl=track, track_initdt=track, max_attempts=96, sum_order=sum_order)
    r32 = o32.forward(x, p)
    assert r32["rc"] == 0
    se = o32.steps_ext()
    dtp, acc = se[:, 2].copy(), se[:, 4].astype(np.int32)
    lam = 100.0   # reference mnist_node.jl:65 (lambda_0), agg = mean (:69)
    svbar = np.full(len(r32["saveval"]), lam / len(r32["saveval"]), np.float32) if track else None
    g32 = o32.backward(ubar, svbar)
    o64 = Oracle(arch, np.float64, TOL, TOL, reg_kind=1, track_ctrl=track, track_initdt=track, max_attempts=96)
    o64.set_replay(dtp, acc)
    r64 = o64.forward(x, p)
    g64 = o64.backward(ubar, svbar)
    out = dict(p=p, x=x, ubar=ubar, dtp=dtp, acc=acc, r32=r32, se32=se, g32=g32, r64=r64, se64=o64.steps_ext(), g64=g64, svbar=svbar)
    _cache[key] = out
    return out


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(np.asarray(b, np.float64)).max())


@pytest.mark.parametrize("persist", [0, -1], ids=["one-launch", "seven-launch"])
def test_replay_headline_trajectory_and_exact_gradient(persist):
    """B = 512, tol 1.4e-8, along the fp32 oracle's step sequence: u_end, and the gradient of <ubar, u_end> with the controller
    and initial-step tracking off (nothing in it depends on the error estimate), against the fp64 oracle."""
    from tests.util import Node
    R = _oracle_runs(B512, 11, 0)
    node = Node(_cfg(B512, regularize=1, track_ctrl=0, track_initdt=0, persist=persist))
    assert node.L.rnde_node_launches_per_attempt(node.h) == (1 if persist == 0 else 7)
    got = node.forward_replay(R["x"], R["p"], R["dtp"], R["acc"], keep_tape=True)
    assert got["nattempts"] == len(R["dtp"]) and got["nfe"] == R["r32"]["nfe"]
    # same fp32 additions => the same (t, dt) sequence, bit for bit
    assert np.array_equal(got["steps"][:, 0], R["se32"][:, 0]) and np.array_equal(got["steps"][:, 1], R["se32"][:, 1])
    assert np.array_equal(got["steps"][:, 3].astype(np.int32), R["acc"])
    e_dev, e_o32 = _rel(got["u"], R["r64"]["u"]), _rel(R["r32"]["u"], R["r64"]["u"])
    print(f"u_end vs fp64 oracle: device {e_dev:.2e}, fp32 oracle {e_o32:.2e}")
    assert e_dev <= 2e-6
    xb, pb, tsb = node.backward(R["ubar"], None)
    ex, ep = _rel(xb, R["g64"][0]), _rel(pb, R["g64"][1])
    ex32, ep32 = _rel(R["g32"][0], R["g64"][0]), _rel(R["g32"][1], R["g64"][1])
    print(f"exact-path gradient vs fp64 oracle: x_bar device {ex:.2e} (fp32 oracle {ex32:.2e}), p_bar device {ep:.2e} (fp32 oracle {ep32:.2e})")
    assert ex <= 2e-5 and ep <= 2e-5
    # tspan cotangents: sums over all D*B entries of O(1e-3) terms
    ts64 = np.asarray(R["g64"][2], np.float64)
    assert np.abs(tsb - ts64).max() <= 1e-4 * max(1.0, np.abs(ts64).max())
    node.close()


@pytest.mark.parametrize("persist", [0, -1], ids=["one-launch", "seven-launch"])
def test_replay_headline_regularised_step(persist, monkeypatch):
    monkeypatch.setenv("RNDE_X3", "0")      # the noise-defined parts are compared with the fp32 oracle's: the kernels that form their products as that oracle models them (matrix mode 0)
    """The full training-step gradient of the headline configuration (cotangent on u_end AND lambda/n on every saved EEst*dt,
    controller and initial step differentiated) along the fp32 oracle's sequence.  EEst is rounding noise here, so the
    parts of the gradient that pass through it differ between ANY two fp32 implementations; the bound is the fp32
    oracle's own distance to the fp64 oracle."""
    from tests.util import Node
    R = _oracle_runs(B512, 11, 1)
    node = Node(_cfg(B512, regularize=1, persist=persist))
    got = node.forward_replay(R["x"], R["p"], R["dtp"], R["acc"], keep_tape=True)
    assert got["nattempts"] == len(R["dtp"]) and got["nfe"] == R["r32"]["nfe"]
    assert _rel(got["u"], R["r64"]["u"]) <= 2e-6
    # per-attempt error estimate: both are the rounding noise of the k_i (EEst = dt |sum btilde_i k_i| / 1.4e-8 with the true sum
    # ~1e-4 of the noise).  The oracle accumulates the K = 785 dot products of layer 1 sequentially, the device as 7 split-K
    # partial sums of MFMA chains, so the device's k_i carry less rounding error and its floor sits at a CONSTANT fraction of
    # the oracle's (measured 0.456-0.479 over the 40 attempts): a stable ratio is what identical noise processes at different
    # amplitudes look like.  Asserted: the ratio's spread is < 10 % of its mean, and its mean lies in [0.3, 1.1].
    ratio = got["steps"][:, 2] / R["se32"][:, 3]
    print("EEst device / fp32 oracle per attempt: min %.3f mean %.3f max %.3f; fp64 oracle EEst max %.1e" % (ratio.min(), ratio.mean(), ratio.max(), R["se64"][:, 3].max()))
    assert 0.3 <= ratio.mean() <= 1.1 and ratio.max() - ratio.min() <= 0.1 * ratio.mean()
    assert len(got["saveval"]) == len(R["r32"]["saveval"])
    sv_ratio = got["saveval"][1:] / R["r32"]["saveval"][1:]
    assert got["saveval"][0] == 0.0 and 0.3 <= sv_ratio.min() and sv_ratio.max() <= 1.1
    reg_dev, reg_o32 = 100.0 * got["saveval"].mean(), 100.0 * R["r32"]["saveval"].mean()
    print(f"regulariser term lambda*mean(saveval): device {reg_dev:.4f}, fp32 oracle {reg_o32:.4f}")
    xb, pb, tsb = node.backward(R["ubar"], R["svbar"])
    assert np.isfinite(xb).all() and np.isfinite(pb).all()

    def dist(a, b):   # relative L2 distance
        a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    dx, dp = dist(xb, R["g64"][0]), dist(pb, R["g64"][1])
    sx, sp = dist(R["g32"][0], R["g64"][0]), dist(R["g32"][1], R["g64"][1])
    print(f"full gradient, relative L2 distance to the fp64 oracle: x_bar device {dx:.3e} / fp32 oracle {sx:.3e}; p_bar device {dp:.3e} / fp32 oracle {sp:.3e}")
    assert dx <= 3.0 * sx + 1e-4 and dp <= 3.0 * sp + 1e-4
    node.close()


def test_vanilla_b64_matches_oracle():
    """BASELINE config 1 shape: vanilla NODE (regularize = false -> the {false,false} call method, reference
    src/models/neural_ode.jl:48-77), B = 64, tol 1.4e-8.  Natural runs: both solutions are accurate to fp32 rounding, so u_end
    agrees whatever step sequence each one chose; replayed along the oracle's sequence it agrees too."""
    from tests.util import Node, Oracle
    arch, p, x, _ = _problem(64, 5)
    o32 = Oracle(arch, np.float32, TOL, TOL, reg_kind=0, max_attempts=96)
    o64 = Oracle(arch, np.float64, TOL, TOL, reg_kind=0, max_attempts=96)
    r32, r64 = o32.forward(x, p), o64.forward(x, p)
    node = Node(_cfg(64, regularize=0))
    got = node.forward(x, p)
    assert len(got["saveval"]) == 0 and len(r32["saveval"]) == 0
    assert got["nfe"] == 3 + 6 * got["nattempts"]
    print(f"vanilla B=64: attempts device {got['nattempts']}, fp32 oracle {r32['nattempts']}, fp64 oracle {r64['nattempts']}; "
          f"u_end vs fp64: device {_rel(got['u'], r64['u']):.2e}, fp32 oracle {_rel(r32['u'], r64['u']):.2e}")
    assert _rel(got["u"], r64["u"]) <= 5e-6
    # (the default matrix mode -- bf16x3 on the matrix cores -- rounds the Dense layers more accurately than the sequential fp32 oracle and takes ~0.6 of its attempts;
    #  the fp32-input-MFMA kernels ~0.73, the fp64 oracle 0.25)
    assert r64["nattempts"] <= got["nattempts"] <= 1.15 * r32["nattempts"]
    se = o32.steps_ext()
    rep = node.forward_replay(x, p, se[:, 2], se[:, 4])
    assert rep["nfe"] == r32["nfe"] and np.array_equal(rep["steps"][:, 1], se[:, 1])
    assert _rel(rep["u"], r64["u"]) <= 5e-6
    node.close()


def test_attempt_count_distribution_at_reference_tolerance(monkeypatch):
    monkeypatch.setenv("RNDE_X3", "0")      # device = device-order oracle, attempt for attempt: matrix mode 0 (fp32-input MFMA); mode 1 takes fewer, tests/test_gpu_x3.py
    """NFE 'parity' at 1.4e-8 is a distribution (DESIGN.md 3.1): 16 seeds, B = 64, device vs fp32 oracle (vs fp64 for scale).
    Measured: device 30.0 +- 0.0, fp32 oracle 40.8 +- 0.6, fp64 oracle ~10: the count is set by the rounding error of the fp32
    GEMMs (the device's error floor is 0.46 of the oracle's, test above, and dt ~ EEst^-0.14/0.2...), not by the ODE; both
    distributions are narrow.  Asserted: the device's mean lies within [0.6, 1.05] of the fp32 oracle's, its spread is no wider
    than 2x + 1, and no seed needs more attempts than 1.15x the oracle's."""
    from tests.util import Node, Oracle
    node = Node(_cfg(64, regularize=1))
    dev, o32, o64 = [], [], []
    for seed in range(16):
        arch, p, x, _ = _problem(64, 100 + seed)
        a = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, max_attempts=96).forward(x, p)
        b = Oracle(arch, np.float64, TOL, TOL, reg_kind=1, max_attempts=96).forward(x, p)
        g = node.forward(x, p)
        dev.append(g["nattempts"]); o32.append(a["nattempts"]); o64.append(b["nattempts"])
        assert g["nattempts"] <= 1.15 * a["nattempts"] + 1
        o3 = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, max_attempts=96, sum_order=3).forward(x, p)     # the oracle in the device's order:
        assert abs(g["nattempts"] - o3["nattempts"]) <= 1                                                   # the SAME count (round 3)
    dev, o32, o64 = np.array(dev, float), np.array(o32, float), np.array(o64, float)
    print(f"attempts over 16 seeds (B=64, tol 1.4e-8): device {dev.mean():.1f} +- {dev.std():.1f} (NFE {3 + 6 * dev.mean():.0f}), "
          f"fp32 oracle {o32.mean():.1f} +- {o32.std():.1f} (NFE {3 + 6 * o32.mean():.0f}), fp64 oracle {o64.mean():.1f} +- {o64.std():.1f}")
    assert 0.6 * o32.mean() <= dev.mean() <= 1.05 * o32.mean()
    assert dev.std() <= 2.0 * o32.std() + 1.0
    node.close()


def test_f_evaluation_equals_the_device_order_oracle_almost_bit_for_bit():
    """One f evaluation, MNIST shape: against the sequential-k oracle the device differs in nearly every entry (rounding), against
    the oracle in the device's order it is bit-identical in >= 90 % of the entries and within 1 ulp of a value < 1 in the rest
    (measured 95 %, 1.19e-7; sequential: 6 %, 4.7e-7).  This is the statement 'the GEMMs are summed as the oracle says'."""
    from tests.util import Node, Oracle
    arch, p, x, _ = _problem(64, 3)
    node = Node(_cfg(64, regularize=1))
    fd = node.feval(x, p, 0.37)
    f0 = Oracle(arch, np.float32, TOL, TOL, sum_order=0).f_eval(p, x, 0.37)
    f1 = Oracle(arch, np.float32, TOL, TOL, sum_order=1).f_eval(p, x, 0.37)
    f3 = Oracle(arch, np.float32, TOL, TOL, sum_order=3).f_eval(p, x, 0.37)
    eq0, eq1, eq3 = float(np.mean(fd == f0)), float(np.mean(fd == f1)), float(np.mean(fd == f3))
    print(f"f evaluation, entries bit-equal to the device: sequential-k oracle {eq0:.3f}, device-order GEMMs {eq1:.3f}, + device tanh {eq3:.3f}; "
          f"max |diff| {np.abs(fd - f0).max():.2e} / {np.abs(fd - f1).max():.2e} / {np.abs(fd - f3).max():.2e}")
    assert eq3 >= 0.90 and np.abs(fd - f3).max() <= 1.2e-7
    assert eq3 > eq1 > eq0
    node.close()


def test_natural_run_attempts_equal_the_device_order_oracle(monkeypatch):
    monkeypatch.setenv("RNDE_X3", "0")      # (matrix mode 0: the oracle's device-order mode mirrors the fp32-input MFMA)
    """NFE parity at the reference tolerance as an EQUALITY (north star: 'trajectories and NFE counts within a stated fp32
    tolerance on identical inputs'): natural runs, device vs the oracle in the device's order, B = 64 over 16 seeds and B = 512
    over 2 -- the same number of attempts (+-1 allowed, 0 observed), the same accept/reject pattern, step sizes within 15 %
    (B = 64; 0.3 % observed at B = 512), per-attempt EEst within 12 % (B = 64) / 2 % (B = 512), u_end to 3e-6."""
    from tests.util import Node, Oracle
    for B, seeds, dt_tol, ee_tol in ((64, range(100, 116), 0.15, 0.12), (512, (11, 12), 0.01, 0.02)):
        node = Node(_cfg(B, regularize=1))
        worst_dt = worst_ee = 0.0
        for seed in seeds:
            arch, p, x, _ = _problem(B, seed)
            o3 = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, max_attempts=96, sum_order=3)
            r3 = o3.forward(x, p)
            se = o3.steps_ext()
            g = node.forward(x, p)
            assert abs(g["nattempts"] - r3["nattempts"]) <= 1, (B, seed, g["nattempts"], r3["nattempts"])
            assert g["nfe"] == 3 + 6 * g["nattempts"]
            n = min(g["nattempts"], r3["nattempts"])
            assert np.array_equal(g["steps"][:n, 3].astype(np.int32), se[:n, 4].astype(np.int32))      # accept / reject pattern
            worst_dt = max(worst_dt, float(np.abs(g["steps"][:n, 1] / se[:n, 1] - 1).max()))
            worst_ee = max(worst_ee, float(np.abs(g["steps"][:n, 2] / se[:n, 3] - 1).max()))
            assert _rel(g["u"], r3["u"]) <= 3e-6
            assert abs(g["saveval"].sum() / r3["saveval"].sum() - 1) <= 0.02
        print(f"B = {B}: attempts equal over {len(list(seeds))} seeds; worst |dt_dev / dt_oracle - 1| {worst_dt:.2e}, worst |EEst ratio - 1| {worst_ee:.2e}")
        assert worst_dt <= dt_tol and worst_ee <= ee_tol
        node.close()


def test_replay_eest_matches_device_order_oracle(monkeypatch):
    monkeypatch.setenv("RNDE_X3", "0")      # (matrix mode 0: the oracle's device-order mode mirrors the fp32-input MFMA)
    """Replay along the device-order oracle's own sequence, B = 512: per-attempt EEst device / oracle in [0.9, 1.1] (VERDICT r02 item 2;
    measured 0.994..1.006), saved values and the regulariser term likewise, and the full training-step gradient (cotangent on u_end and
    lambda / n on every EEst * dt, everything tracked) now agrees with THAT oracle's gradient far better than either agrees with fp64 --
    the noise-defined part of the gradient is the same noise."""
    from tests.util import Node
    R = _oracle_runs(B512, 11, 1, sum_order=3)
    node = Node(_cfg(B512, regularize=1))
    got = node.forward_replay(R["x"], R["p"], R["dtp"], R["acc"], keep_tape=True)
    assert got["nattempts"] == len(R["dtp"]) and got["nfe"] == R["r32"]["nfe"]
    ratio = got["steps"][:, 2] / R["se32"][:, 3]
    print("EEst device / device-order oracle per attempt: min %.4f mean %.4f max %.4f" % (ratio.min(), ratio.mean(), ratio.max()))
    assert 0.9 <= ratio.min() and ratio.max() <= 1.1
    assert got["saveval"][0] == 0.0
    sv_ratio = got["saveval"][1:] / R["r32"]["saveval"][1:]
    assert 0.9 <= sv_ratio.min() and sv_ratio.max() <= 1.1
    reg_dev, reg_o = 100.0 * got["saveval"].mean(), 100.0 * R["r32"]["saveval"].mean()
    assert abs(reg_dev / reg_o - 1) <= 0.02
    xb, pb, tsb = node.backward(R["ubar"], R["svbar"])

    def dist(a, b):
        a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    dx, dp = dist(xb, R["g32"][0]), dist(pb, R["g32"][1])
    sx, sp = dist(R["g32"][0], R["g64"][0]), dist(R["g32"][1], R["g64"][1])
    print(f"full gradient, relative L2: device vs device-order oracle x_bar {dx:.3e} p_bar {dp:.3e}; that oracle vs fp64 x_bar {sx:.3e} p_bar {sp:.3e}; "
          f"regulariser term device {reg_dev:.4f} oracle {reg_o:.4f}")
    # measured: p_bar 1.7e-3 from the device-order oracle against that oracle's 1.0e-2 from fp64 (6x closer); x_bar 5.6e-4 against 6.7e-4
    # (the gradient THROUGH rounding noise is itself noise: 0.6 % agreement of EEst does not make its derivative agree)
    assert dp <= 0.5 * sp + 1e-4 and dx <= 1.5 * sx + 1e-4
    node.close()
