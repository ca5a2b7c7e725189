"""GPU parity: reverse pass (x-bar, p-bar, tspan-bar) of the HIP path against the CPU oracle.

Only truncation-dominated regimes are compared element-wise (identical accept/reject sequences, see
test_gpu_forward.py); tolerance: 2e-3 of the largest gradient entry (fp32 accumulation over
~10^2 evaluations x batch in a different association order), fp64 oracle as arbiter."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (the two test_node cases at tol 1e-2 / 3e-2 contain a rejected step)
CASES = [("test_node", 7, 1e-3, 3.0, 1.0, 3, 16), ("small", 20, 1e-3, 4.0, 1.0, 3, 16), ("mnist", 32, 1e-3, 3.0, 1.0, 3, 16),
         ("test_node", 3, 1e-2, 5.0, 2.0, 18, 16), ("mnist", 19, 1e-2, 6.0, 2.0, 5, 16), ("test_node", 3, 3e-2, 5.0, 2.0, 27, 16),
         ("small", 9, 1e-3, 4.0, 1.0, 4, 16), ("mnist", 12, 1e-3, 3.0, 1.0, 6, 16), ("mnist", 37, 1e-3, 3.0, 1.0, 7, 16), ("small", 33, 1e-3, 4.0, 1.0, 8, 16)]


@pytest.mark.parametrize("kind,B,tol,scale,t1,seed,col_tile", CASES)
@pytest.mark.parametrize("wu,ws", [(1.0, 0.0), (1.0, 50.0)])
def test_backward_matches_oracle(kind, B, tol, scale, t1, seed, col_tile, wu, ws):
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, seed, scale)
    rng = np.random.default_rng(100 + seed)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=col_tile))
    got = node.forward(x, p, 0.0, t1, keep_tape=True)
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1)
    r32 = o32.forward(x, p, 0.0, t1)
    r64 = o64.forward(x, p, 0.0, t1)
    assert (got["steps"][:, 3] == r32["steps"][:, 3]).all() and got["nattempts"] == r32["nattempts"]
    ubar = (wu * rng.standard_normal(x.shape)).astype(np.float32)
    svbar = np.full(len(got["saveval"]), ws, dtype=np.float32)
    xb, pb, tsb = node.backward(ubar, svbar)
    xb32, pb32, tsb32 = o32.backward(ubar, svbar)
    same64 = r64["nattempts"] == r32["nattempts"] and (r64["steps"][:, 3] == r32["steps"][:, 3]).all()
    print(f"{kind} B={B} natt={got['nattempts']} nrej={int((got['steps'][:,3]==0).sum())} "
          f"x-bar err {rel_err(xb, xb32):.2e}  p-bar err {rel_err(pb, pb32):.2e}  tspan {tsb} vs {tsb32}")
    # conditioning of the case itself: how far the fp32 oracle is from the fp64 oracle (rejected steps only
    # occur in rough regimes where gradients are sensitive; those cases are judged relative to that spread)
    cx = cp = 0.0
    if same64:
        xb64, pb64, tsb64 = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
        cx, cp = rel_err(xb32, xb64), rel_err(pb32, pb64)
        print(f"   oracle f32 vs f64 spread: x {cx:.2e} p {cp:.2e}; device vs f64: x {rel_err(xb, xb64):.2e} p {rel_err(pb, pb64):.2e}")
        # cases with a rejected step live in a rough regime (|tspan-bar| up to 5e3): 5 % there, tight elsewhere
        slack = 5e-2 if (got["steps"][:, 3] == 0).any() else 0.0
        assert rel_err(xb, xb64) <= 2e-3 + 3 * cx + slack
        assert rel_err(pb, pb64) <= 2e-3 + 3 * cp + slack
    slack = 5e-2 if (got["steps"][:, 3] == 0).any() else 0.0
    assert rel_err(xb, xb32) <= 2e-3 + 4 * cx + slack
    assert rel_err(pb, pb32) <= 2e-3 + 4 * cp + slack
    assert np.abs(tsb - tsb32).max() <= (2e-3 + 4 * max(cx, cp) + slack) * max(1.0, np.abs(tsb32).max())


@pytest.mark.parametrize("B", [37, 64])
def test_weight_gradient_kernels_agree(B, monkeypatch):
    """p-bar of the MNIST form through the weight-gradient GEMM variants: the 16x16x4 kernel with the wide side split in two
    halves (rnde_wgrad3_kernel, the default for this shape) with and without the launches that run on a second stream underneath
    the sweep (RNDE_WGRAD_SIDE: per cent of the attempts; needs >= 8 attempts, hence the tighter tolerance here), and the staged
    32x32x2 kernel (RNDE_WGRAD_V2).  Same evaluations, different summation order over (evaluation, column): 1e-5 of the largest
    entry.  The direct-from-global kernel (RNDE_WGRAD_LEGACY) serves the odd shapes of the oracle comparisons above."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node, rel_err
    arch, p, x = _setup("mnist", B, 11, 3.0)
    rng = np.random.default_rng(5)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    out = {}
    for name, env in (("v3", {"RNDE_WGRAD_SIDE": "0"}), ("v3_side", {"RNDE_WGRAD_SIDE": "50"}), ("v3_side_all", {"RNDE_WGRAD_SIDE": "100"}),
                      ("v2", {"RNDE_WGRAD_V2": "1"})):
        for k in ("RNDE_WGRAD_V2", "RNDE_WGRAD_SIDE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        node = Node(_cfg(arch, B, reltol=1e-5, abstol=1e-5, col_tile=16))
        got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
        assert got["nattempts"] >= 8
        out[name] = node.backward(ubar, np.full(len(got["saveval"]), 10.0, dtype=np.float32))
    for name in ("v3_side", "v3_side_all", "v2"):
        assert np.array_equal(out["v3"][0], out[name][0]), name            # x-bar does not involve the GEMM
        assert rel_err(out["v3"][1], out[name][1]) <= 1e-5, name
    assert np.abs(out["v3"][1]).max() > 0


def test_weight_gradient_kernels_agree_on_another_shape(monkeypatch):
    """The same comparison on a shape that is not the headline's: D = 720 (45 row tiles: halves of 23 and 22), H = 64, B = 40 (48 padded
    columns: every second 32-column step of an evaluation is half empty) -- the LDS-DMA staging of rnde_wgrad3_kernel (units, masked lanes,
    synthetic rows, stale columns) against the staged 32x32x2 kernel and against the fp64 oracle."""
    from tests.test_gpu_forward import _cfg
    from tests.util import Node, Oracle, arch_mnist, glorot_params, rel_err
    rng = np.random.default_rng(3)
    arch = arch_mnist(720, 64)
    p = glorot_params(arch, rng, np.float32, 3.0)
    B = 40
    x = rng.uniform(0, 1, (B, 720)).astype(np.float32)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    out = {}
    for name, env in (("v3", {"RNDE_WGRAD_SIDE": "0"}), ("v3_side", {"RNDE_WGRAD_SIDE": "100"}), ("v2", {"RNDE_WGRAD_V2": "1"})):
        for k in ("RNDE_WGRAD_V2", "RNDE_WGRAD_SIDE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        node = Node(_cfg(arch, B, reltol=1e-3, abstol=1e-3, col_tile=16))
        got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
        out[name] = node.backward(ubar, np.full(len(got["saveval"]), 10.0, dtype=np.float32))
    for name in ("v3_side", "v2"):
        assert rel_err(out["v3"][1], out[name][1]) <= 1e-5, name
    o64 = Oracle(arch, np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r64 = o64.forward(x, p, 0.0, 1.0)
    if r64["nattempts"] == got["nattempts"]:
        xb64, pb64, _ = o64.backward(ubar.astype(np.float64), np.full(len(got["saveval"]), 10.0))
        assert rel_err(out["v3"][1], pb64) <= 2e-3 and rel_err(out["v3"][0], xb64) <= 2e-3


@pytest.mark.parametrize("t1", [2.1, 2.2, 3.4, 3.0, 4.8])
def test_reverse_pass_of_a_long_solve(monkeypatch, t1):
    """More than ~65 attempts on the stage engine with the side-stream launches: the slab bookkeeping of the weight-gradient GEMMs (side
    launches of 16 chunks each + 127 chunks behind the sweep) must hold, where a stale capacity check of the legacy kernels' chunk count
    used to refuse the reverse pass ("weight-gradient slab overflow" -- met in a training run whose step count had grown)."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node, rel_err
    arch, p, x = _setup("mnist", 32, 21, 2.0)
    rng = np.random.default_rng(9)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    out = {}
    for name, env in (("v3", {}), ("v2", {"RNDE_WGRAD_V2": "1"})):
        monkeypatch.delenv("RNDE_WGRAD_V2", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        node = Node(_cfg(arch, 32, reltol=1.4e-8, abstol=1.4e-8, max_attempts=256))
        got = node.forward(x, p, 0.0, t1, keep_tape=True)
        print("attempts", got["nattempts"])      # (the refused window was 70..79 and 110..119 attempts)
        assert got["nattempts"] >= 45, got["nattempts"]      # (t1 = 3.0 / 3.4 / 4.8 reach the windows with the default matrix mode, which takes ~0.75 of the fp32-MFMA attempts)
        out[name] = node.backward(ubar, np.full(len(got["saveval"]), 1.0, dtype=np.float32))
    assert np.isfinite(out["v3"][1]).all() and np.abs(out["v3"][1]).max() > 0
    # (v3 runs on the matrix cores in the default matrix mode -- rnde_wgradx.h, six bf16 cross products per fp32 product -- and v2 on the fp32-input MFMA: two
    #  roundings of the same sums, each ~1e-6 from the exact value over K = 32 columns x ~300 evaluations)
    assert rel_err(out["v3"][1], out["v2"][1]) <= 1e-4


def test_backward_requires_tape():
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    from regneuralde_jl_amd._lib import RndeError
    arch, p, x = _setup("small", 4, 0, 1.0)
    node = Node(_cfg(arch, 4, reltol=1e-3, abstol=1e-3))
    node.forward(x, p, keep_tape=False)
    with pytest.raises(RndeError):
        node.backward(np.ones_like(x))


@pytest.mark.parametrize("kind,B,tol,scale,seed", [("test_node", 5, 1e-3, 3.0, 3), ("small", 12, 1e-3, 4.0, 4), ("mnist", 17, 1e-3, 3.0, 5)])
@pytest.mark.parametrize("reg,agg", [(2, "max"), (2, "mean"), (3, "mean"), (4, "mean")])
def test_stiffness_regulariser_matches_oracle(kind, B, tol, scale, seed, reg, agg):
    """regularize = stiff_est / error_stiff_est (experiments/mnist_node.jl:70-99, AutoTsit5(Tsit5()) semantics) and |eigen_est * dt| (the
    reference's own test, test/test_node.jl:75,:84): callback values and the reverse pass of eigen_est = ||k7-k6|| / ||u-g6|| against the oracle."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node, Oracle, rel_err
    arch, p, x = _setup(kind, B, seed, scale)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, regularize=reg, col_tile=16))
    got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
    o32 = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=reg)
    o64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=reg)
    r32, r64 = o32.forward(x, p), o64.forward(x, p)
    assert got["nattempts"] == r32["nattempts"] and (got["steps"][:, 3] == r32["steps"][:, 3]).all()
    np.testing.assert_allclose(got["saveval"], r64["saveval"], rtol=2e-2, atol=1e-5)
    rng = np.random.default_rng(7)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    sv = got["saveval"]
    svbar = np.zeros(len(sv), dtype=np.float32)
    if agg == "max":
        svbar[int(np.argmax(r64["saveval"]))] = 2.0
    else:
        svbar[:] = 2.0 / len(sv)
    xb, pb, tsb = node.backward(ubar, svbar)
    xb64, pb64, _ = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    xb32, pb32, _ = o32.backward(ubar, svbar)
    cx, cp = rel_err(xb32, xb64), rel_err(pb32, pb64)
    print(f"reg {reg}/{agg} {kind}: x-bar {rel_err(xb, xb64):.2e} (oracle f32 {cx:.2e})  p-bar {rel_err(pb, pb64):.2e} (oracle f32 {cp:.2e})")
    assert rel_err(xb, xb64) <= 3e-3 + 3 * cx
    assert rel_err(pb, pb64) <= 3e-3 + 3 * cp


def test_stiffness_gradient_at_trained_like_weights_is_bounded():
    """Round-5 review, item 7 (reference experiments/mnist_node.jl:70-81: `stiff_est`, lambda 0.1, `maximum`).  The Glorot-init test above says little about
    where the regulariser acts.  After 24, 48 and 72 optimiser steps of the reference loop on the synthetic set (tools/stiff_grad_trained.py, default matrix mode):
    d(lambda * max_n |eigen_est_n| / 3.5068) / dp of the device against the fp64 oracle replaying the device's own step sequence, per parameter block (largest
    deviation over the block's largest fp64 entry), with the oracle's own fp32 build beside it -- what ANY fp32 evaluation of this term is worth in that state.
    The run is chaotic (which states it passes through changes with any kernel detail), and in states where the largest estimate barely exceeds the initial
    constant the term is NOISE in fp32: the fp32 restatement itself is 0.4 .. 5 x off there and the device 0.3 .. 60 x (profiles/r06_stiff_grad_trained.json);
    nothing can be asserted about such a state but that the numbers are finite.
    STATED BOUND: in every state that RESOLVES the term in fp32 -- the fp32 restatement within 2e-2 of fp64 on every block -- the device is within 5e-2 on every
    block (measured 2e-3 .. 3e-2 against the restatement's 5e-3 .. 1.4e-2) and its cosine with the fp64 gradient is >= 0.999; at least one of the three states
    resolves it (measured: the 48-step one; after 72 steps the restatement is 2e-2 .. 9e-2 off and the device 4e-3 .. 1.3e-2)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("stiff_grad_trained", os.path.join(root, "tools", "stiff_grad_trained.py"))
    sg = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(sg)
    resolved = 0
    blocks = ("W1", "b1", "W2", "b2")
    for steps, p2, x in sg.weights_after((24, 48, 72)):
        r = sg.reg_gradient_check(p2, x, 600)
        dev, o32 = r["device_vs_fp64"], r["oracle_f32_vs_fp64"]
        print(f"after {steps} steps: attempts {r['attempts']}, max saved value {r['saveval_max_fp64']:.3f} (initial constant {r['init_value']:.3f}); device vs fp64 "
              + "  ".join(f"{k} {v['rel_max']:.2e}" for k, v in dev.items()) + " | fp32 oracle vs fp64 " + "  ".join(f"{k} {v['rel_max']:.2e}" for k, v in o32.items())
              + f" | cos {dev['all']['cos']:.6f}")
        assert all(np.isfinite(dev[n]["rel_max"]) for n in blocks)
        if max(o32[n]["rel_max"] for n in blocks) <= 2e-2:
            resolved += 1
            assert max(dev[n]["rel_max"] for n in blocks) <= 5e-2 and dev["all"]["cos"] >= 0.999, (steps, dev)
    assert resolved >= 1


def test_retired_column_owner_tiles_are_refused():
    """col_tile 4 / 8 selected the round-1 column-owner engine (retired in round 4, tools/experiments/column_owner/): creation says so."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    from regneuralde_jl_amd._lib import RndeError
    arch, p, x = _setup("small", 4, 0, 1.0)
    for tile in (4, 8):
        with pytest.raises(RndeError) as e:
            Node(_cfg(arch, 4, regularize=2, col_tile=tile))
        assert "retired" in str(e.value)


@pytest.mark.parametrize("kind,B,tol,scale,saveat,reg", [("test_node", 5, 1e-3, 3.0, np.linspace(0, 1, 7), 1), ("small", 12, 1e-3, 4.0, np.array([0.1, 0.5, 0.9]), 1),
                                                          ("mnist", 19, 1e-3, 3.0, np.linspace(0, 1, 13), 1), ("small", 33, 1e-4, 4.0, np.array([0.0, 0.25, 1.0]), 0),
                                                          ("small", 7, 1e-3, 4.0, np.array([0.31, 0.32, 0.33, 0.34]), 3)])
def test_saveat_reverse_matches_oracle(kind, B, tol, scale, saveat, reg):
    """Reverse pass of the {R,true} call methods (neural_ode.jl:79-108,:146-180): the cotangent is the D x T x B array
    Tracker hands back for diffeqsol_to_3dtrackedarray (src/utils.jl:17-19); dense-output points carry their theta
    dependence on t and dt."""
    from tests.util import Node, Oracle, rel_err
    from tests.test_gpu_forward import _setup, _cfg
    arch, p, x = _setup(kind, B, 5, scale)
    sa = saveat.astype(np.float32)
    o = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=reg, track_ctrl=1, track_initdt=1)
    ref = o.forward(x, p, saveat=sa)
    rng = np.random.default_rng(3)
    ubar = rng.standard_normal(ref["u"].shape).astype(np.float32)
    svbar = (rng.standard_normal(len(ref["saveval"])) * 10).astype(np.float32) if reg else None
    rx, rp, rt = o.backward(ubar, svbar)
    n = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16, regularize=reg, track_ctrl=1, track_initdt=1))
    got = n.forward_saveat(x, p, sa, keep_tape=True)
    assert got["nfe"] == ref["nfe"] and len(got["saveval"]) == len(ref["saveval"])
    gx, gp, gt = n.backward(ubar, svbar)
    assert rel_err(gx, rx) < 2e-3
    assert rel_err(gp, rp) < 2e-3
    assert np.abs(gt - rt).max() <= 2e-3 * max(1.0, np.abs(rt).max())
