"""The one-launch forward solve with its Dense-layer products on the MATRIX CORES (csrc/rnde_x3.h, include/rnde.h: rnde_node_set_matrix_mode):
both operands split exactly into three bf16 numbers, the six leading cross products as v_mfma_f32_16x16x32_bf16, fp32 accumulation.

The oracle's device-order mode mirrors the fp32-input MFMA (a k-ordered FMA chain); nothing mirrors a bf16 matrix instruction's inner sum bit for
bit, so this mode's parity is stated against the fp64 restatement replayed along the device's OWN step sequence, with the fp32-MFMA kernel's distance to
the same fp64 run beside it -- the tolerance every fp32 implementation of the reference's solve has to meet (reference call site:
`solve(prob, Tsit5(); ...)`, src/models/neural_ode.jl:131-137; dynamics experiments/mnist_node.jl:41-54).  Tolerances, all relative to the largest entry:
    u_end                 <= 2e-6   (the bound of tests/test_gpu_replay.py for the fp32-MFMA kernel)
    exact-path gradients  <= 2e-5   (cotangent on u_end only, controller and initial-step tracking off)
    regulariser noise     the gradient of lambda * mean(EEst*dt) at Glorot weights (rounding noise in either mode): no larger than the fp32-MFMA kernel's
and the property that motivates the mode: at the reference tolerance (1.4e-8: step size set by rounding noise) it takes NO MORE attempted steps than
the fp32-MFMA kernel on any of the seeds, and fewer in the mean."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1.4e-8


def _problem(B, seed, scale=1.0):
    from tests.util import arch_mnist, glorot_params
    rng = np.random.default_rng(seed)
    arch = arch_mnist()
    p = glorot_params(arch, rng, np.float32, scale)
    x = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ubar = (rng.standard_normal((B, 784)) / B).astype(np.float32)
    return arch, p, x, ubar


def _cfg(B, tol=TOL, **kw):
    from tests.util import make_cfg
    return make_cfg([784, 100, 784], ["tanh", "tanh"], B, reltol=tol, abstol=tol, max_attempts=200, **kw)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(np.asarray(b, np.float64)).max())


def test_matrix_mode_is_the_default_where_it_applies_and_can_be_switched():
    from tests.util import Node, make_cfg
    from regneuralde_jl_amd import _lib
    node = Node(_cfg(64))
    assert node.L.rnde_node_matrix_mode(node.h) == 1                      # headline geometry: bf16x3 by default
    assert node.L.rnde_node_set_matrix_mode(node.h, 0) == 0 and node.L.rnde_node_matrix_mode(node.h) == 0
    assert node.L.rnde_node_set_matrix_mode(node.h, 1) == 0 and node.L.rnde_node_matrix_mode(node.h) == 1
    assert node.L.rnde_node_set_matrix_mode(node.h, 7) == _lib.BAD_ARG
    node.close()
    small = Node(make_cfg([36, 10, 36], ["tanh", "tanh"], 16, reltol=1e-3, abstol=1e-3))      # another geometry: the fp32 kernels serve it
    assert small.L.rnde_node_matrix_mode(small.h) == 0
    assert small.L.rnde_node_set_matrix_mode(small.h, 1) == 0 and small.L.rnde_node_matrix_mode(small.h) == 0      # asked for, not available: stays 0 and says so
    small.close()


@pytest.mark.parametrize("B,scale,seed", [(512, 1.0, 11), (512, 2.5, 12), (200, 1.0, 13), (16, 3.0, 14)])
def test_x3_solve_against_the_fp64_restatement(B, scale, seed):
    """Natural run at the reference tolerance; the fp64 oracle replays the device's own (dt, accept) sequence."""
    from tests.util import Node, Oracle
    arch, p, x, ubar = _problem(B, seed, scale)
    out = {}
    for mode in (0, 1):
        node = Node(_cfg(B, regularize=1, track_ctrl=0, track_initdt=0), matrix_mode=mode)
        got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
        assert node.L.rnde_node_one_launch_solves(node.h) == 1 and node.L.rnde_node_matrix_mode(node.h) == mode
        st = got["steps"]
        o64 = Oracle(arch, np.float64, TOL, TOL, reg_kind=1, track_ctrl=0, track_initdt=0, max_attempts=200)
        o64.set_replay(st[:, 1].astype(np.float64), st[:, 3].astype(np.int32))
        r64 = o64.forward(x.astype(np.float64), p.astype(np.float64))
        assert r64["rc"] == 0 and got["nfe"] == r64["nfe"] == 3 + 6 * len(st)
        # the same (t, dt) sequence: t advances by the device's fp32 additions
        assert np.abs(st[:, 0] - o64.steps_ext()[:, 0]).max() <= 1e-6
        xb, pb, tsb = node.backward(ubar, None)
        g64 = o64.backward(ubar.astype(np.float64), None)
        out[mode] = dict(att=len(st), eu=_rel(got["u"], r64["u"]), ex=_rel(xb, g64[0]), ep=_rel(pb, g64[1]),
                         sv=float(np.abs(got["saveval"] - r64["saveval"]).max()), svmax=float(np.abs(got["saveval"]).max()))
        node.close()
    print(f"B {B} scale {scale}: attempts fp32-MFMA {out[0]['att']} bf16x3 {out[1]['att']} | u_end vs fp64 {out[0]['eu']:.2e} / {out[1]['eu']:.2e} | "
          f"x-bar {out[0]['ex']:.2e} / {out[1]['ex']:.2e} | p-bar {out[0]['ep']:.2e} / {out[1]['ep']:.2e} | max saved EEst*dt {out[0]['svmax']:.2e} / {out[1]['svmax']:.2e}")
    assert out[1]["eu"] <= 2e-6 and out[1]["ex"] <= 2e-5 and out[1]["ep"] <= 2e-5
    assert out[1]["att"] <= out[0]["att"]


def test_x3_takes_fewer_attempts_than_the_fp32_mfma_kernel_over_seeds():
    """16 seeds at B = 64 (the statistic of tests/test_gpu_replay.py): attempts of the two matrix modes beside the oracles' (fp32 sequential 40.8, fp32 in
    the fp32-MFMA order 30.0 = the fp32-MFMA kernel, fp64 10)."""
    from tests.util import Node
    att = {0: [], 1: []}
    for seed in range(16):
        arch, p, x, _ = _problem(64, 100 + seed)
        for mode in (0, 1):
            node = Node(_cfg(64, regularize=1), matrix_mode=mode)
            att[mode].append(node.forward(x, p)["nattempts"])
            node.close()
    m0, m1 = float(np.mean(att[0])), float(np.mean(att[1]))
    print(f"attempts over 16 seeds: fp32-MFMA {m0:.1f} (min {min(att[0])}, max {max(att[0])}), bf16x3 {m1:.1f} (min {min(att[1])}, max {max(att[1])})")
    assert all(b <= a for a, b in zip(att[0], att[1])) and m1 <= 0.9 * m0


def test_x3_solve_is_deterministic_and_survives_reuse():
    """Two handles, repeated solves, another batch width in between: the same bits every time (no atomics, no uninitialised operand image)."""
    from tests.util import Node
    arch, p, x, ubar = _problem(512, 21)
    a, b = Node(_cfg(512, regularize=1), matrix_mode=1), Node(_cfg(512, regularize=1), matrix_mode=1)
    r0 = a.forward(x, p, keep_tape=True)
    g0 = a.backward(ubar, None)
    a.forward(x[:100], p)                    # another tile count in between
    r1 = a.forward(x, p, keep_tape=True)
    g1 = a.backward(ubar, None)
    r2 = b.forward(x, p, keep_tape=True)
    for r in (r1, r2):
        assert np.array_equal(r0["u"], r["u"]) and np.array_equal(r0["steps"], r["steps"]) and np.array_equal(r0["saveval"], r["saveval"])
    assert all(np.array_equal(u, v) for u, v in zip(g0, g1))
    a.close(); b.close()


def test_x3_regulariser_noise_gradient_is_no_larger_than_the_fp32_kernels():
    """At Glorot-initial weights and the reference tolerance the saved EEst*dt are rounding noise, and so is the gradient of lambda * mean(EEst*dt)
    (lambda = 100, experiments/mnist_node.jl:65): the reverse pass pushes a cotangent of norm ~ eb / (sqrt(N) sk) ~ 1e4 along the NOISE direction
    through Sum_j btilde_j J_j^T, which cancels to the true O(dt^4) sensitivity only up to fp32 rounding.  What a training step sees of it is a
    perturbation of ~1 % of the signal gradient in either matrix mode (measured: |g_reg| 0.12 with the fp32-input MFMA, 0.067 with bf16x3, against
    |g_signal| 9.4; the fp32 oracles replaying the same steps: 0.04 - 0.44).  Bound: bf16x3's is no larger than 1.5 x the fp32-MFMA kernel's, and
    both stay under 5 % of the signal's norm.  (Comparing this term with the fp64 oracle's is meaningless: along these steps ITS EEst is 1e-4 of
    the device's, and so is everything that is divided by it.)"""
    from tests.util import Node
    arch, p, x, ubar = _problem(512, 31)
    d = {}
    for mode in (0, 1):
        node = Node(_cfg(512, regularize=1), matrix_mode=mode)
        got = node.forward(x, p, keep_tape=True)
        n = len(got["saveval"])
        _, g_reg, _ = node.backward(np.zeros_like(ubar), np.full(n, 100.0 / n, np.float32))
        node.forward(x, p, keep_tape=True)
        _, g_sig, _ = node.backward(ubar, None)
        d[mode] = (float(np.linalg.norm(g_reg)), float(np.linalg.norm(g_sig)), got["nattempts"])
        node.close()
    print(f"|g_reg| / |g_signal|: fp32-MFMA {d[0][0]:.3e} / {d[0][1]:.3e} ({d[0][2]} attempts) | bf16x3 {d[1][0]:.3e} / {d[1][1]:.3e} ({d[1][2]} attempts)")
    assert d[1][0] <= 1.5 * d[0][0] and d[1][0] <= 0.05 * d[1][1] and d[0][0] <= 0.05 * d[0][1]


@pytest.mark.parametrize("B,side", [(512, 30), (512, 0), (100, 100), (37, 0), (1500, 30)])
def test_x3_weight_gradient_forms_agree(B, side, monkeypatch):
    """The parameter gradient of matrix mode 1 through its two weight-gradient kernels -- the quarter form with two LDS images (rnde_wgrad4x_kernel, the default:
    16-byte operand loads, wave-uniform operand per wave, repeated units in the idle lanes, two register sets) and the single-buffered half form
    (rnde_wgrad3x_kernel, RNDE_X3_WGRAD_HALF=1) -- and through the fp32-input-MFMA kernel (RNDE_X3_WGRAD_OFF=1): the same evaluations off the same tape (the
    forward solve and the reverse sweep do not depend on the switch: x-bar bit-equal), the six-term products in the same order per 32 columns, chunked
    differently: 5e-6 of the largest entry between the two x3 forms and 3e-5 against the fp32 kernel for the signal's gradient; with the error-estimate regulariser's
    cotangent on top (saved values that are rounding noise at this tolerance, divided by it: cancelling sums 1e4 x the result) 3e-3.  Batches that are no multiple of 32 (half-empty last step
    of every evaluation), with and without the launches underneath the sweep, and a batch above 1024 (two-tile forward kernel, launch-per-attempt reverse)."""
    from tests.util import Node, rel_err
    arch, p, x, ubar = _problem(B, 41)
    out = {}
    for name, env in (("quarter", {}), ("half", {"RNDE_X3_WGRAD_HALF": "1"}), ("fp32", {"RNDE_X3_WGRAD_OFF": "1"})):
        for k in ("RNDE_X3_WGRAD_HALF", "RNDE_X3_WGRAD_OFF"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("RNDE_WGRAD_SIDE", str(side))
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        node = Node(_cfg(B, regularize=1), matrix_mode=1)
        got = node.forward(x, p, keep_tape=True)
        n = len(got["saveval"])
        sig = node.backward(ubar, np.zeros(n, np.float32))                       # the signal's gradient alone
        node.forward(x, p, keep_tape=True)
        reg = node.backward(ubar, np.full(n, 10.0 / n, np.float32))             # + the error-estimate regulariser's (rounding noise amplified ~1e4 x, see above)
        out[name] = (sig, reg)
        node.close()
    for name in ("half", "fp32"):
        assert np.array_equal(out["quarter"][0][0], out[name][0][0]) and np.array_equal(out["quarter"][1][0], out[name][1][0]), name
    e = {(a, k): rel_err(out["quarter"][k][1], out[a][k][1]) for a in ("half", "fp32") for k in (0, 1)}
    print(f"B = {B}, side {side}: p-bar quarter vs half {e['half', 0]:.2e} (with the regulariser's cotangent {e['half', 1]:.2e}), quarter vs fp32 kernel {e['fp32', 0]:.2e} ({e['fp32', 1]:.2e})")
    assert np.abs(out["quarter"][0][1]).max() > 0
    assert e["half", 0] <= 5e-6 and e["fp32", 0] <= 3e-5
    assert e["half", 1] <= 3e-3 and e["fp32", 1] <= 3e-3      # (sums of cancelling terms 1e4 x the result: the summation order shows at 1e-4 .. 1e-3 of it)


@pytest.mark.parametrize("B,tol,scale,reg", [(512, TOL, 1.0, 1), (64, TOL, 1.0, 1), (200, 1e-3, 3.0, 1), (37, 1e-4, 2.0, 0), (1024, TOL, 1.0, 1)])
def test_x3_one_launch_solve_is_bit_identical_to_the_x3_attempt_kernel(B, tol, scale, reg, monkeypatch):
    """The two x3 kernels issue the same matrix instructions in the same order on the same operands: the one-launch solve (B <= 512) and one launch per
    attempted step (RNDE_STAGE_SOLVE=0; also what saveat, the coupled controller, B > 512 and the fallback run) agree bit for bit -- steps, u_end, saved
    values, and the gradients of the reverse pass that follows (the mirror of tests/test_gpu_solve.py for matrix mode 1)."""
    from tests.util import Node
    arch, p, x, ubar = _problem(B, 41, scale)
    out = []
    for solve in ("1", "0"):
        monkeypatch.setenv("RNDE_STAGE_SOLVE", solve)
        node = Node(_cfg(B, tol, regularize=reg), matrix_mode=1)
        got = node.forward(x, p, keep_tape=True)
        assert node.L.rnde_node_one_launch_solves(node.h) == (1 if solve == "1" and B <= 512 else 0)
        n = len(got["saveval"])
        g = node.backward(ubar, np.full(n, 100.0 / max(n, 1), np.float32) if n else None)
        out.append((got, g))
        node.close()
    (a, ga), (b, gb) = out
    assert a["nattempts"] == b["nattempts"] and np.array_equal(a["steps"], b["steps"])
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["saveval"], b["saveval"])
    assert all(np.array_equal(u, v) for u, v in zip(ga, gb))


def test_x3_attempt_kernel_serves_saveat_against_the_fp64_restatement():
    """saveat runs launch-per-attempt: in matrix mode 1 on the x3 attempt kernel.  Dense-output states vs the fp64 oracle along the device's steps."""
    from tests.util import Node, Oracle
    arch, p, x, _ = _problem(96, 51, 2.0)
    sa = np.linspace(0.1, 0.9, 5).astype(np.float32)      # (inside (0, 1): the fp64 replay's last t is the SUM of the device's fp32 step sizes, 1 - O(1e-8): a save time at exactly 1 would stay unfilled there)
    node = Node(_cfg(96, 1e-4, regularize=1), matrix_mode=1)
    got = node.forward_saveat(x, p, sa, keep_tape=False)
    st = got["steps"]
    o64 = Oracle(arch, np.float64, 1e-4, 1e-4, reg_kind=1, max_attempts=200)
    o64.set_replay(st[:, 1].astype(np.float64), st[:, 3].astype(np.int32))
    r64 = o64.forward(x.astype(np.float64), p.astype(np.float64), saveat=sa.astype(np.float64))
    assert got["u"].shape == r64["u"].shape and _rel(got["u"], r64["u"]) <= 5e-6
    node.close()


# ---------------------------------------------------------------- the oracle's MIRROR of matrix mode 1 (Oracle(sum_order=7), oracle/rnde_oracle.c f_col_stage_x3)
def test_x3_attempt_equals_the_oracles_mirror_almost_bit_for_bit():
    """One attempted step (six f evaluations) from the same (uprev, k1, t, dt): the device in matrix mode 1 against the oracle restating it -- operands split
    into three bf16 numbers, six cross products per fp32 product, every v_mfma_f32_16x16x32_bf16 as four exact 8-term sums added with a rounding each (the model
    tools/micro/mfma_bf16_numerics.hip fits to raw matrix-core output: 96.6 % of single instructions bit-equal), four accumulators added smallest first, the device's tanh.
    Over the SIX chained evaluations of an attempt: >= 45 % of the k entries bit-equal (measured 52.5 %; the oracle in the fp32-MFMA order: 14 %), every entry within
    1 ulp of a value < 1 (1.2e-7), u_new bit-equal in >= 85 % of its entries (92 %), EEst within 5 %."""
    from tests.util import Node, Oracle
    arch, p, x, _ = _problem(64, 3)
    rng = np.random.default_rng(8)
    k1 = np.tanh(rng.standard_normal((64, 784))).astype(np.float32)
    node = Node(_cfg(64, regularize=1), matrix_mode=1)
    kd, ud, ed = node.attempt(x, k1, p, 0.2, 0.03)
    res = {}
    for so in (3, 7):
        ko, uo, eo, _ = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, sum_order=so).attempt(p, x, k1, 0.2, 0.03)
        res[so] = (float(np.mean(kd == ko)), float(np.abs(kd - ko).max()), float(np.mean(ud == uo)), eo)
    print(f"attempt, k entries bit-equal to the device (matrix mode 1): oracle in the fp32-MFMA order {res[3][0]:.3f} (max diff {res[3][1]:.2e}), oracle mirror of mode 1 "
          f"{res[7][0]:.3f} (max diff {res[7][1]:.2e}); u_new bit-equal {res[7][2]:.3f}; EEst device {ed:.4f} mirror {res[7][3]:.4f} fp32-order {res[3][3]:.4f}")
    assert res[7][0] >= 0.45 and res[7][1] <= 1.5e-7 and res[7][0] > res[3][0] + 0.3 and res[7][2] >= 0.85
    assert abs(ed / res[7][3] - 1) <= 0.05
    node.close()


def test_x3_natural_run_attempts_equal_the_oracles_mirror():
    """NFE parity of the DEFAULT matrix mode at the reference tolerance as an equality: natural runs, device (mode 1) vs Oracle(sum_order=7), B = 64 over 16
    seeds and B = 512 over 2 -- the same number of attempts (+-1 allowed), the same accept / reject pattern, step sizes and per-attempt EEst close, u_end to 3e-6
    (the statement tests/test_gpu_replay.py::test_natural_run_attempts_equal_the_device_order_oracle makes for mode 0)."""
    from tests.util import Node, Oracle
    for B, seeds, dt_tol, ee_tol in ((64, range(100, 116), 0.25, 0.25), (512, (11, 12), 0.03, 0.06)):
        node = Node(_cfg(B, regularize=1), matrix_mode=1)
        worst_dt = worst_ee = 0.0
        diff = []
        for seed in seeds:
            arch, p, x, _ = _problem(B, seed)
            o7 = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, max_attempts=200, sum_order=7)
            r7 = o7.forward(x, p)
            se = o7.steps_ext()
            g = node.forward(x, p)
            diff.append(g["nattempts"] - r7["nattempts"])
            assert abs(diff[-1]) <= 1, (B, seed, g["nattempts"], r7["nattempts"])
            n = min(g["nattempts"], r7["nattempts"])
            assert np.array_equal(g["steps"][:n, 3].astype(np.int32), se[:n, 4].astype(np.int32))
            worst_dt = max(worst_dt, float(np.abs(g["steps"][:n, 1] / se[:n, 1] - 1).max()))
            worst_ee = max(worst_ee, float(np.abs(g["steps"][:n, 2] / se[:n, 3] - 1).max()))
            assert _rel(g["u"], r7["u"]) <= 3e-6
        print(f"B = {B}: device (matrix mode 1) - oracle mirror attempts over {len(diff)} seeds: {diff}; worst |dt ratio - 1| {worst_dt:.2e}, worst |EEst ratio - 1| {worst_ee:.2e}")
        assert worst_dt <= dt_tol and worst_ee <= ee_tol
        node.close()
