"""GPU parity tests of the stochastic layer (rnde_nsde_*, reference src/models/neural_sde.jl) against the CPU oracle
(oracle/rnde_sde_oracle.c) on identical inputs AND identical noise (the pool of standard normals both consume in order).

fp32 tolerances, stated per test.  At the reference tolerance (reltol = abstol = 0.14, experiments/mnist_nsde.jl:79-80) the
error estimate is truncation dominated by orders of magnitude, so -- unlike the ODE at 1.4e-8 -- the accept/reject sequence,
the number of attempts and the number of noise draws must match the oracle EXACTLY, and dt to 1e-5 relative (powf differs
by an ulp between the device and libm).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AGGR = dict(qmax=10.0, gamma=1.0, beta2=1e-9)   # a controller that rejects often (the recalled defaults hardly ever do)


@pytest.fixture(autouse=True, params=["1", "0"], ids=["multi-wave", "one-wave"])
def _solve_kernel(request, monkeypatch):
    """The reference's shape has two whole-solve kernels (csrc/rnde_sdemw.h: four waves per tile, the default; csrc/rnde_sde.h: one
    wave per tile, also what any other shape runs): every test of this file runs on both (RNDE_SDE_MW is read at handle creation)."""
    monkeypatch.setenv("RNDE_SDE_MW", request.param)
    yield


def _nets(kind):
    from oracle.oracle import make_arch
    from oracle.oracle_sde import arch_nsde_diffusion, arch_nsde_drift
    if kind == "nsde":          # experiments/mnist_nsde.jl:73-74
        return arch_nsde_drift(), arch_nsde_diffusion()
    if kind == "small":
        return make_arch([3, 5, 3], ["tanh", "identity"], False), make_arch([3, 3], ["identity"], False)
    if kind == "deep":          # three-layer drift, two-layer tanh diffusion, D = 20 (another k-step count)
        return make_arch([20, 33, 17, 20], ["tanh", "tanh", "identity"], False), make_arch([20, 9, 20], ["tanh", "identity"], False)
    if kind == "wide":          # D = 40 -> 16 k-step registers per array
        return make_arch([40, 64, 40], ["tanh", "identity"], False), make_arch([40, 40], ["tanh"], False)
    raise KeyError(kind)


def _setup(kind, B, seed, n_pool, scale=2.0, dscale=0.5):
    from oracle.oracle_sde import nsde_params
    drift, diff = _nets(kind)
    rng = np.random.default_rng(seed)
    p = nsde_params(drift, diff, rng, np.float32, scale, dscale)
    p = (p + 0.05 * rng.standard_normal(len(p))).astype(np.float32)
    D = drift.dims[0]
    x = rng.standard_normal((B, D)).astype(np.float32)
    noise = rng.standard_normal((n_pool, 2, B, D)).astype(np.float32)
    return drift, diff, p, x, noise


def _cfg(drift, diff, B, **kw):
    from tests.util import make_nsde_cfg
    dd = [drift.dims[i] for i in range(drift.n_layers + 1)]
    da = ["tanh" if drift.act[i] else "identity" for i in range(drift.n_layers)]
    gd = [diff.dims[i] for i in range(diff.n_layers + 1)]
    ga = ["tanh" if diff.act[i] else "identity" for i in range(diff.n_layers)]
    return make_nsde_cfg(dd, da, gd, ga, B, **kw)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("kind,B,solver", [("nsde", 16, "SOSRI"), ("nsde", 37, "SOSRI"), ("small", 5, "SOSRI"), ("small", 1, "SRIW1"),
                                           ("deep", 21, "SOSRI2"), ("wide", 19, "SOSRI")])
def test_attempt_matches_oracle(kind, B, solver):
    """One attempted step from a given (uprev, dt, dW, dZ): the eight stage values, the proposed state and the error estimate.
    Differences: association order of the fp32 dot products (MFMA chains vs sequential) and 1-ulp tanh -> 2e-5 abs on O(1) values."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup(kind, B, 3, 1)
    dt = 0.07
    dW, dZ = np.sqrt(dt) * noise[0, 0], np.sqrt(dt) * noise[0, 1]
    o32 = SdeOracle(drift, diff, np.float32, tableau=solver)
    o64 = SdeOracle(drift, diff, np.float64, tableau=solver)
    kg_ref, un_ref, e_ref = o32.attempt(p, x, dt, dW, dZ)
    _, un64, e64 = o64.attempt(p, x, dt, dW, dZ)
    node = NsdeNode(_cfg(drift, diff, B, solver=solver))
    kg, un, e = node.attempt(x, p, dt, dW, dZ)
    assert np.abs(kg - kg_ref).max() <= 2e-5 * max(1.0, np.abs(kg_ref).max())
    assert np.abs(un - un64).max() <= 2e-5 * max(1.0, np.abs(un64).max())
    assert abs(e - e64) <= 2e-5 * e64 + 1e-6 and abs(e_ref - e64) <= 2e-5 * e64 + 1e-6
    node.close()


@pytest.mark.parametrize("kind,B,tol,ctrl,solver,scale,dscale,replay", [
    ("nsde", 64, 0.14, {}, "SOSRI", 2.0, 0.5, False), ("nsde", 37, 0.05, AGGR, "SOSRI", 3.0, 1.5, True), ("small", 7, 0.05, AGGR, "SOSRI", 2.0, 0.5, False),
    ("deep", 21, 0.05, AGGR, "SOSRI2", 2.5, 0.8, True), ("small", 3, 0.05, AGGR, "SRIW1", 2.0, 0.5, False), ("wide", 33, 0.03, AGGR, "SOSRI", 3.0, 1.0, False)])
def test_solve_matches_oracle_on_the_same_noise(kind, B, tol, ctrl, solver, scale, dscale, replay):
    """Whole solve: same accept/reject sequence, same number of attempts and of noise draws, NFE counters, saved values and
    u(t1) (2e-4 relative: the trajectory amplifies stage-level rounding over ~60 steps).  The cases with hundreds of attempts and
    an oscillating controller (two thirds rejected) are chaotic -- one borderline EEst flips an accept and the sequences part --
    so those follow the oracle's sequence through rnde_nsde_forward_replay: the rejection bookkeeping (stack moves, bridges,
    draws) is then compared step for step."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup(kind, B, 5, 400, scale, dscale)
    o = SdeOracle(drift, diff, np.float32, tol, tol, tableau=solver, max_attempts=399, **ctrl)
    ref = o.forward(x, p, noise)
    assert ref["rc"] == 0
    node = NsdeNode(_cfg(drift, diff, B, reltol=tol, abstol=tol, solver=solver, max_attempts=399, **ctrl))
    got = node.forward(x, p, noise, replay=np.stack([ref["steps"][:, 1], ref["steps"][:, 3]], 1) if replay else None)
    if ctrl:
        assert (ref["steps"][:, 3] == 0).sum() >= 3, "this case is meant to exercise rejections"
    assert got["nattempts"] == ref["nattempts"] and np.array_equal(got["steps"][:, 3], ref["steps"][:, 3])
    assert got["ndraws"] == ref["ndraws"]
    assert got["nfe1"] == ref["nfe1"] == 2 + 4 * ref["nattempts"] and got["nfe2"] == ref["nfe2"]
    # (the last step is t1 - t, and t is a sum of ~60 steps that each agree to ~1e-6 relative: 2e-6 absolute)
    assert np.allclose(got["steps"][:, 1], ref["steps"][:, 1], rtol=2e-5, atol=2e-6) and np.allclose(got["steps"][:, 0], ref["steps"][:, 0], rtol=2e-5, atol=2e-6)
    assert np.allclose(got["steps"][:, 2], ref["steps"][:, 2], rtol=5e-4, atol=1e-6)
    assert len(got["saveval"]) == len(ref["saveval"]) and np.allclose(got["saveval"], ref["saveval"], rtol=5e-4, atol=1e-7)
    assert _rel(got["u"], ref["u"]) <= 2e-4
    node.close()


@pytest.mark.parametrize("kind,B,tol,ctrl,scale,dscale", [("nsde", 48, 0.14, {}, 2.0, 0.5), ("nsde", 21, 0.05, AGGR, 3.0, 1.5), ("small", 6, 0.05, AGGR, 2.0, 0.5),
                                                          ("deep", 17, 0.05, AGGR, 2.5, 0.8)])
def test_reverse_pass_matches_oracle(kind, B, tol, ctrl, scale, dscale):
    """x_bar, p_bar for cotangents on u(t1) and on every saved EEst*dt, against the fp32 and fp64 oracle along the same
    (replayed) sequence.  Bound: 2e-3 of the largest entry plus the case's own fp32-vs-fp64 spread."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup(kind, B, 9, 400, scale, dscale)
    rng = np.random.default_rng(1)
    o32 = SdeOracle(drift, diff, np.float32, tol, tol, max_attempts=399, **ctrl)
    r32 = o32.forward(x, p, noise)
    ubar = rng.standard_normal(x.shape).astype(np.float32) / B
    svbar = (rng.standard_normal(len(r32["saveval"])) * 0.3).astype(np.float32)
    g32 = o32.backward(ubar, svbar)
    o64 = SdeOracle(drift, diff, np.float64, tol, tol, max_attempts=399, **ctrl)
    o64.set_replay(r32["steps"][:, 1], r32["steps"][:, 3].astype(np.int32))
    r64 = o64.forward(x, p, noise)
    assert r64["nattempts"] == r32["nattempts"]
    g64 = o64.backward(ubar, svbar)
    node = NsdeNode(_cfg(drift, diff, B, reltol=tol, abstol=tol, max_attempts=399, **ctrl))
    got = node.forward(x, p, noise, keep_tape=True)
    assert np.array_equal(got["steps"][:, 3], r32["steps"][:, 3])
    if ctrl:
        assert (got["steps"][:, 3] == 0).sum() >= 3, "this case is meant to include rejected steps"
    xb, pb = node.backward(ubar, svbar)
    for name, a, b32, b64 in (("x_bar", xb, g32[0], g64[0]), ("p_bar", pb, g32[1], g64[1])):
        scale = np.abs(b64).max()
        spread = np.abs(np.asarray(b32, np.float64) - b64).max()
        err = np.abs(np.asarray(a, np.float64) - b64).max()
        print(f"{kind} B={B} {name}: device-fp64 {err / scale:.2e}, fp32 oracle-fp64 {spread / scale:.2e}")
        assert err <= 2e-3 * scale + 3 * spread, (name, err, scale, spread)
    # drift and diffusion gradients both populated
    assert np.abs(pb[:node.len]).max() > 0 and np.abs(pb[node.len:]).max() > 0
    node.close()


def test_config5_full_size_vs_fp64_oracle():
    """BASELINE config 5: D = 32, drift 32 -> 64 -> 32, diffusion 32 -> 32, B = 512, trajectories = 1, tol 0.14 (reference
    experiments/mnist_nsde.jl:70-84,:96), forward + reverse with lambda = 10 on mean(saveval) (:47-48, :99)."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    B = 512
    drift, diff, p, x, noise = _setup("nsde", B, 21, 257, scale=1.0, dscale=1.0)
    o64 = SdeOracle(drift, diff, np.float64, max_attempts=256)
    r64 = o64.forward(x, p, noise)
    assert r64["rc"] == 0
    node = NsdeNode(_cfg(drift, diff, B, max_attempts=256))
    got = node.forward(x, p, noise, keep_tape=True)
    print(f"config 5: attempts {got['nattempts']} (fp64 oracle {r64['nattempts']}), nfe1 = nfe2 = {got['nfe1']}, draws {got['ndraws']}")
    assert got["nattempts"] == r64["nattempts"] and np.array_equal(got["steps"][:, 3], r64["steps"][:, 3])
    assert got["nfe1"] == r64["nfe1"] and got["nfe2"] == r64["nfe2"] and got["ndraws"] == r64["ndraws"]
    assert _rel(got["u"], r64["u"]) <= 2e-4
    assert np.allclose(got["saveval"], r64["saveval"], rtol=5e-4, atol=1e-7)
    rng = np.random.default_rng(2)
    ubar = (rng.standard_normal(x.shape) / B).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 10.0 / len(got["saveval"]), np.float32)
    xb, pb = node.backward(ubar, svbar)
    g64 = o64.backward(ubar, svbar)
    ex, ep = _rel(xb, g64[0]), _rel(pb, g64[1])
    print(f"config 5 reverse vs fp64 oracle: x_bar {ex:.2e}, p_bar {ep:.2e}")
    assert ex <= 1e-3 and ep <= 1e-3
    node.close()


def test_replay_and_fixed_step_mode():
    """rnde_nsde_forward_replay: a fixed-step run (all accepted, equal dt) and a replayed adaptive sequence reproduce the oracle."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup("nsde", 32, 13, 80)
    n = 20
    o = SdeOracle(drift, diff, np.float32, max_attempts=79)
    o.set_replay(np.full(n, 1.0 / n, np.float32), np.ones(n, np.int32))
    ref = o.forward(x, p, noise)
    node = NsdeNode(_cfg(drift, diff, 32, max_attempts=79))
    got = node.forward(x, p, noise, replay=np.stack([np.full(n, 1.0 / n), np.ones(n)], 1))
    assert got["nattempts"] == n == ref["nattempts"] and got["ndraws"] == ref["ndraws"] == n
    assert _rel(got["u"], ref["u"]) <= 1e-4
    node.close()


def test_library_noise_stream_is_standard_normal_and_reproducible():
    """noise_dev = NULL: the library's Philox4x32-10 + Box-Muller stream.  Moments of 4M samples within 5 sigma, no
    correlation between neighbours, the same seed gives the same solve and another seed another one."""
    import ctypes as C
    import torch
    from regneuralde_jl_amd import _lib
    from tests.util import NsdeNode
    L = _lib.lib()
    n = 1 << 22
    buf = torch.empty(n, dtype=torch.float32, device="cuda")
    assert L.rnde_normal_fill(buf.data_ptr(), n, 1234, 7, None) == 0
    torch.cuda.synchronize()
    z = buf.cpu().numpy().astype(np.float64)
    s = 5.0 / np.sqrt(n)
    assert abs(z.mean()) < s and abs(z.var() - 1) < s * np.sqrt(2) and abs((z ** 3).mean()) < s * np.sqrt(15) and abs((z ** 4).mean() - 3) < s * np.sqrt(96)
    assert abs(np.mean(z[:-1] * z[1:])) < s and abs(np.mean(z[:-4:4] * z[2::4][:len(z[:-4:4])])) < 2 * s
    assert np.abs(z).max() < 6.5 and (np.abs(z) > 4).mean() < 2e-4
    buf2 = torch.empty(n, dtype=torch.float32, device="cuda")
    L.rnde_normal_fill(buf2.data_ptr(), n, 1234, 8, None)
    torch.cuda.synchronize()
    assert abs(np.mean(z * buf2.cpu().numpy())) < s      # streams are independent
    drift, diff, p, x, _ = _setup("nsde", 64, 2, 1)
    node = NsdeNode(_cfg(drift, diff, 64))
    a = node.forward(x, p, None, seed=99)
    b = node.forward(x, p, None, seed=99)
    c = node.forward(x, p, None, seed=100)
    assert np.array_equal(a["u"], b["u"]) and a["nattempts"] == b["nattempts"]
    assert not np.array_equal(a["u"], c["u"])
    assert a["nfe1"] == 2 + 4 * a["nattempts"] == a["nfe2"]
    node.close()


def test_shortened_noise_pool_is_refilled_on_demand():
    """Library noise: the pool is filled for 1.5 x the draws of the LAST solve; a solve that needs more is redone with the whole pool.  Draw k
    depends on (seed, k) alone, so a long solve behind a short one on the same handle must equal the same solve on a fresh handle, bit for bit."""
    from tests.util import NsdeNode
    drift, diff, p, x, _ = _setup("nsde", 48, 3, 1)
    fresh = NsdeNode(_cfg(drift, diff, 48, reltol=0.14, abstol=0.14))
    ref = fresh.forward(x, p, None, seed=7, t1=1.0)
    fresh.close()
    node = NsdeNode(_cfg(drift, diff, 48, reltol=0.14, abstol=0.14))
    short = node.forward(x, p, None, seed=7, t1=0.02)          # a handful of attempts: the next pool is small
    assert short["nattempts"] + 16 < ref["nattempts"] * 2 // 3, (short["nattempts"], ref["nattempts"])
    again = node.forward(x, p, None, seed=7, t1=1.0)           # runs out of it, is redone
    assert again["nattempts"] == ref["nattempts"] and np.array_equal(again["u"], ref["u"])
    node.close()


def test_error_paths():
    from regneuralde_jl_amd import _lib
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup("nsde", 16, 1, 4)
    node = NsdeNode(_cfg(drift, diff, 16, max_attempts=100))
    r = node.forward(x, p, noise, check=False)                 # 4 draws cannot cover ~60 attempts
    assert r["rc"] == _lib.BAD_ARG and b"noise pool" in node.L.rnde_nsde_last_error(node.h)
    with pytest.raises(_lib.RndeError):
        node.backward(np.zeros_like(x))                        # no tape
    node2 = NsdeNode(_cfg(drift, diff, 16, max_attempts=5))
    r = node2.forward(x, p, np.random.default_rng(0).standard_normal((8, 2, 16, 32)).astype(np.float32), check=False)
    assert r["rc"] == _lib.MAX_ATTEMPTS
    node.close(); node2.close()


# ---- the layer mirror (regneuralde.jl_amd/nsde.py) -------------------------------------------------------------------------
def test_layer_call_contract_and_autograd():
    """TrackedNeuralDSDE(model1, model2, tspan, regularize, SOSRI; kw...)(x, p) -> (u, nfe1, nfe2, sv) (neural_sde.jl:116-146),
    differentiable through torch.autograd; gradients against the fp64 oracle on the same noise pool."""
    import torch
    import regneuralde_jl_amd as rn
    from oracle.oracle_sde import SdeOracle, arch_nsde_diffusion, arch_nsde_drift
    g = torch.Generator().manual_seed(3)
    B = 24
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g),
                                [0.0, 1.0], True, "SOSRI", save_everystep=False, reltol=0.14, abstol=0.14, save_start=False, max_batch=B)
    assert nsde.len == 32 * 64 + 64 + 64 * 32 + 32 and nsde.P == nsde.len + 32 * 32 + 32      # 4,192 + 1,056
    x = torch.randn(B, 32, generator=g).cuda().requires_grad_(True)
    p = (nsde.p * 2.0).cuda().requires_grad_(True)
    noise = torch.randn(257, 2, B, 32, generator=g).cuda()
    u, nfe1, nfe2, sv = nsde(x, p, func="error_est", noise=noise)
    assert u.shape == (B, 32) and nfe1 == nfe2 and (nfe1 - 2) % 4 == 0 and sv.saveval[0] == 0
    w = torch.randn(B, 32, generator=g).cuda() / B
    loss = (u * w).sum() + 10.0 * sv.saveval.mean()
    loss.backward()
    o = SdeOracle(arch_nsde_drift(), arch_nsde_diffusion(), np.float64, max_attempts=256)
    r = o.forward(x.detach().cpu().numpy(), p.detach().cpu().numpy(), noise.cpu().numpy())
    assert r["nfe1"] == nfe1 and _rel(u.detach().cpu().numpy(), r["u"]) <= 2e-4
    xb, pb = o.backward(w.cpu().numpy(), np.full(len(r["saveval"]), 10.0 / len(r["saveval"])))
    assert _rel(x.grad.cpu().numpy(), xb) <= 1e-3 and _rel(p.grad.cpu().numpy(), pb) <= 1e-3
    # unregularised method: sv is nothing (neural_sde.jl:64-82)
    plain = rn.TrackedNeuralDSDE(nsde.model1, nsde.model2, [0.0, 1.0], False, "SOSRI", reltol=0.14, abstol=0.14, max_batch=B)
    with torch.no_grad():
        u2, a, b, sv2 = plain(x.detach(), p.detach(), noise=noise)
    assert sv2 is None and a == nfe1 and torch.allclose(u2, u.detach(), rtol=0, atol=0)
    # library noise: advancing seed -> a different path every call, as a fresh Julia RNG draw would be
    with torch.no_grad():
        v1 = plain(x.detach(), p.detach())[0]
        v2 = plain(x.detach(), p.detach())[0]
    assert not torch.equal(v1, v2)


def test_classifier_nsde_trajectories():
    """ClassifierNSDE (supervised_classification.jl:82-103): _expand repeats the batch, the logits are averaged over the
    trajectories; with trajectories = 1 the call equals presde -> nsde -> postsde."""
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(5)
    B, T = 8, 3
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g),
                                [0.0, 1.0], True, "SOSRI", reltol=0.14, abstol=0.14, max_batch=B * T)
    model = rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde, rn.Dense(32, 10, "identity", g))
    assert [p.numel() for p in model.trainable()] == [25120, 5248, 330]
    x = torch.rand(B, 784, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].cuda()
    noise = torch.randn(257, 2, B * T, 32, generator=g).cuda()
    z, nfe1, nfe2, sv = model(x, trajectories=T, func="error_est", noise=noise)
    assert z.shape == (B, 10)
    # by hand: trajectory t of input b is row t*B + b of the expanded batch
    h = model._dense(x.repeat(T, 1), model.p1, model.pre_shape)
    with torch.no_grad():
        u = nsde(h.detach().contiguous(), model.p2.detach(), func="error_est", noise=noise)[0]
    zz = model._dense(u, model.p3, model.post_shape).reshape(T, B, 10).mean(0)
    assert torch.allclose(z, zz, atol=1e-6)
    loss, ce, reg, a, b = rn.nsde_loss_function(x, y, model, trajectories=1, lam=10.0)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in model.trainable())


@pytest.mark.parametrize("T", [1, 3])
def test_fused_nsde_step_matches_autograd(T):
    """fused_nsde_loss_and_grad (no tape library in the loop) against torch.autograd through ClassifierNSDE: same library noise stream
    (same seed), same loss and the same three gradients."""
    import torch
    import regneuralde_jl_amd as rn
    B = 16

    def make():
        g = torch.Generator().manual_seed(9)
        nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g),
                                    [0.0, 1.0], True, "SOSRI", reltol=0.14, abstol=0.14, max_batch=B * T, seed=77)
        return rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde, rn.Dense(32, 10, "identity", g)), g
    m1, g = make()
    m2, _ = make()
    x = torch.rand(B, 784, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].cuda()
    loss1, ce1, reg1, a1, b1 = rn.nsde_loss_function(x, y, m1, trajectories=T, lam=10.0)
    loss1.backward()
    loss2, ce2, reg2, a2, b2 = rn.fused_nsde_loss_and_grad(m2, x, y, trajectories=T, lam=10.0)
    assert (a1, b1) == (a2, b2)
    assert abs(float(loss1.detach()) - float(loss2)) <= 1e-5 * max(1.0, abs(float(loss1.detach())))
    for p, q in zip(m1.trainable(), m2.trainable()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6 * float(p.grad.abs().max())), float((p.grad - q.grad).abs().max())


def test_one_call_nsde_step_equals_the_three_calls(monkeypatch):
    """rnde_nsde_classifier_grad (solve + head + reverse sweep in one call, the head queued before the forward's host wait) against
    rnde_nsde_forward + rnde_nsde_classifier_head + rnde_nsde_backward_async: same kernels, same noise stream (same seed), same order --
    loss, both NFE counters and the three gradients bit for bit, at the reference's size (B = 512), three steps with changing weights."""
    import torch
    import regneuralde_jl_amd as rn
    B = 512

    def make():
        g = torch.Generator().manual_seed(21)
        nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g),
                                    [0.0, 1.0], True, "SOSRI", reltol=0.14, abstol=0.14, max_batch=B, seed=5)
        return rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde, rn.Dense(32, 10, "identity", g)), g
    m1, g = make()
    m2, _ = make()
    x = torch.rand(B, 784, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].cuda()
    for rep in range(3):
        monkeypatch.setenv("RNDE_ONE_CALL", "0")
        l1, ce1, reg1, a1, b1 = rn.fused_nsde_loss_and_grad(m1, x, y, lam=10.0)
        monkeypatch.setenv("RNDE_ONE_CALL", "1")
        l2, ce2, reg2, a2, b2 = rn.fused_nsde_loss_and_grad(m2, x, y, lam=10.0)
        torch.cuda.synchronize()
        assert (a1, b1) == (a2, b2) and float(ce1) == float(ce2) and reg1 == pytest.approx(reg2, rel=1e-6)
        for p, q in zip(m1.trainable(), m2.trainable()):
            assert torch.equal(p.grad, q.grad)
        with torch.no_grad():
            for p, q in zip(m1.trainable(), m2.trainable()):
                d = 0.01 * torch.randn(p.shape, generator=g).cuda()
                p.add_(d); q.add_(d)


def test_dropped_graph_returns_the_handle():
    """A taped forward whose graph is dropped without backward() must not pin its handle (and tape) for ever -- the reference's
    per-epoch NFE probe `_, nfe, _ = node(dummy)` (experiments/mnist_node.jl:236) runs with tracking on."""
    import gc
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(1)
    dyn = rn.MLPDynamics(36, 10, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", reltol=1e-3, abstol=1e-3, max_batch=8, max_attempts=32)
    x = torch.rand(8, 36, generator=g).cuda()
    p = node.p.cuda().requires_grad_(True)
    for _ in range(12):
        u, nfe, sv = node(x, p)
        del u, sv
        gc.collect()
    assert sum(len(v) for v in node._handles.values()) == 1
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(8, 16, "tanh", g), rn.Dense(16, 8, "identity", g)), rn.Dense(8, 8, "identity", g), [0.0, 1.0], True,
                                reltol=0.14, abstol=0.14, max_batch=8)
    q = nsde.p.cuda().requires_grad_(True)
    xs = torch.randn(8, 8, generator=g).cuda()
    for _ in range(12):
        out = nsde(xs, q, func="error_est")
        del out
        gc.collect()
    assert sum(len(v) for v in nsde._handles.values()) == 1
    # and a float64 parameter vector is refused instead of being reinterpreted
    with pytest.raises(TypeError):
        node(x, p.detach().double())


def test_fixed_shape_kernels_agree_with_the_generic_ones():
    """The reference's own shape (32 -> 64 -> 32, 32 -> 32) runs on kernels with the shapes as compile-time constants; cfg.generic = 1
    keeps the run-time-shape kernels.  Same arithmetic in the same order: same step sequence, states and gradients to rounding."""
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup("nsde", 50, 17, 257)
    rng = np.random.default_rng(3)
    ubar = (rng.standard_normal(x.shape) / 50).astype(np.float32)
    outs = []
    for generic in (0, 1):
        node = NsdeNode(_cfg(drift, diff, 50, generic=generic))
        r = node.forward(x, p, noise, keep_tape=True)
        xb, pb = node.backward(ubar, np.full(len(r["saveval"]), 0.2, np.float32))
        outs.append((r, xb, pb))
        node.close()
    (a, xa, pa), (b, xb, pb) = outs
    assert a["nattempts"] == b["nattempts"] and np.array_equal(a["steps"][:, 3], b["steps"][:, 3])
    assert _rel(a["u"], b["u"]) <= 1e-6 and _rel(xa, xb) <= 1e-5 and _rel(pa, pb) <= 1e-5


@pytest.mark.parametrize("kind,B,saveat", [("nsde", 40, np.linspace(0, 1, 30)), ("small", 7, np.array([0.13, 0.5, 0.77])), ("deep", 21, np.array([0.0, 0.4, 1.0]))])
def test_saveat_methods_match_oracle(kind, B, saveat):
    """The {R,true} call methods (neural_sde.jl:44-61,:84-113; sde_toy_problem.jl saves 30 points): D x T x B result and its reverse."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup(kind, B, 23, 300)
    sa = saveat.astype(np.float32)
    o32 = SdeOracle(drift, diff, np.float32, max_attempts=299)
    o64 = SdeOracle(drift, diff, np.float64, max_attempts=299)
    o32.set_saveat(sa); o64.set_saveat(sa)
    r32, r64 = o32.forward(x, p, noise), o64.forward(x, p, noise)
    assert r32["nattempts"] == r64["nattempts"]
    node = NsdeNode(_cfg(drift, diff, B, max_attempts=299))
    got = node.forward(x, p, noise, saveat=sa, keep_tape=True)
    assert got["u"].shape == (B, len(sa), drift.dims[0]) and got["nattempts"] == r64["nattempts"] and got["nfe1"] == r64["nfe1"]
    assert _rel(got["u"], r64["u"]) <= 2e-4
    rng = np.random.default_rng(4)
    ubar = (rng.standard_normal(got["u"].shape) / B).astype(np.float32)
    svbar = (0.2 * rng.standard_normal(len(got["saveval"]))).astype(np.float32)
    xb, pb = node.backward(ubar, svbar)
    g64 = o64.backward(ubar, svbar)
    assert _rel(xb, g64[0]) <= 1e-3 and _rel(pb, g64[1]) <= 1e-3
    node.close()


def test_layer_saveat_method():
    """TrackedNeuralDSDE{R,true}: saveat= at construction -> (B, T, D) result (Julia D x T x B), differentiable; the toy problem's
    shapes (experiments/sde_toy_problem.jl:11-14,:50-60: D = 2, 100 trajectories of one u0, 30 save points) with a drift the ABI covers."""
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(8)
    ts = torch.linspace(0, 1, 30)
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(2, 50, "tanh", g), rn.Dense(50, 2, "identity", g)), rn.Dense(2, 2, "identity", g), [0.0, 1.0], True, "SOSRI",
                                saveat=ts, reltol=0.3, abstol=0.3, max_batch=100)
    assert nsde.return_multiple
    u0 = torch.tensor([[2.0, 0.0]]).repeat(100, 1).cuda()
    p = nsde.p.cuda().requires_grad_(True)
    sol, nfe1, nfe2, sv = nsde(u0, p, func="error_est")
    assert sol.shape == (100, 30, 2) and torch.equal(sol[:, 0], u0) and nfe1 == nfe2
    means = sol.mean(dim=0)                                  # mean(sol; dims = 3) of the reference's loss
    loss = (means ** 2).mean() + sol.var(dim=0).mean() + 0.2 * sv.saveval.sum()
    loss.backward()
    assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0


def test_layer_save_everystep_method():
    """TrackedNeuralDSDE built with save_everystep = true (neural_sde.jl:14): rnde_nsde_forward_everystep returns the state after every accepted
    step (t0 first); on an explicit noise pool the result and its gradients are those of the saveat call at the same times, bit for bit, and the
    last state is the plain solve's end state."""
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(11)
    nets = lambda: (rn.Chain(rn.Dense(6, 12, "tanh", torch.Generator().manual_seed(11)), rn.Dense(12, 6, "identity", torch.Generator().manual_seed(12))),
                    rn.Dense(6, 6, "identity", torch.Generator().manual_seed(13)))
    B = 20
    kw = dict(reltol=0.05, abstol=0.05, max_batch=B, max_attempts=200)
    every = rn.TrackedNeuralDSDE(*nets(), [0.0, 1.0], True, "SOSRI", save_everystep=True, **kw)
    plain = rn.TrackedNeuralDSDE(*nets(), [0.0, 1.0], True, "SOSRI", **kw)
    x = torch.randn(B, 6, generator=g).cuda().requires_grad_(True)
    p = every.p.cuda().clone().requires_grad_(True)
    noise = torch.randn(300, 2, B, 6, generator=g).cuda()
    sol, nfe1, nfe2, sv = every(x, p, func="error_est", noise=noise)
    ts = every.last_times
    n = len(ts)
    assert sol.shape == (B, n, 6) and n >= 4 and ts[0] == 0.0 and ts[-1] == 1.0 and all(b > a for a, b in zip(ts, ts[1:]))
    assert torch.equal(sol[:, 0], x.detach()) and nfe1 == nfe2
    with torch.no_grad():
        ue, m1, _, _ = plain(x, p, func="error_est", noise=noise)
    assert m1 == nfe1 and torch.equal(sol[:, -1].detach(), ue)
    w = torch.randn(B, n, 6, generator=g).cuda()
    ((sol * w).sum() + 3.0 * sv.saveval.sum()).backward()
    gx, gp = x.grad.clone(), p.grad.clone()
    x.grad = None; p.grad = None
    at = rn.TrackedNeuralDSDE(*nets(), [0.0, 1.0], True, "SOSRI", saveat=ts, **kw)
    sol2, k1, _, sv2 = at(x, p, func="error_est", noise=noise)
    assert k1 == nfe1 and torch.equal(sol2.detach(), sol.detach())
    ((sol2 * w).sum() + 3.0 * sv2.saveval.sum()).backward()
    assert torch.equal(x.grad, gx) and torch.equal(p.grad, gp)


def test_more_than_8192_columns_take_the_one_wave_solve_kernel():
    """The four-waves-per-tile solve kernel needs a workgroup per tile resident (<= 512 tiles, two per CU); a call with 8300 columns switches to
    the one-wave-per-tile kernel (same handle, same tape layout, the reverse sweep stays on the four-wave kernel): the call contract and
    finite, non-trivial gradients."""
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(6)
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g),
                                [0.0, 1.0], True, "SOSRI", reltol=0.14, abstol=0.14, max_batch=8300)
    x = torch.randn(8300, 32, generator=g).cuda().requires_grad_(True)
    p = nsde.p.cuda().clone().requires_grad_(True)
    u, nfe1, nfe2, sv = nsde(x, p, func="error_est")
    assert u.shape == (8300, 32) and torch.isfinite(u).all() and nfe1 == nfe2 and nfe1 % 4 == 2
    (u.sum() + sv.saveval.sum()).backward()
    assert torch.isfinite(x.grad).all() and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0


def test_evaluation_size_5120_columns_vs_oracle():
    """The reference's evaluation call (experiments/mnist_nsde.jl:154-155: accuracy(...; trajectories = 10) on a batch of 512 = 5,120 columns in ONE
    solve with ONE error norm, supervised_classification.jl:87-98): 320 workgroups of the four-wave kernel, two per CU on 64 of them, meeting
    through agent-scope entries.  Same accept / reject sequence, draws and NFE as the fp32 oracle on the same noise; end state and the reverse
    pass against it."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    B = 5120
    drift, diff, p, x, noise = _setup("nsde", B, 23, 200, scale=1.0, dscale=1.0)
    o = SdeOracle(drift, diff, np.float32, 0.14, 0.14, max_attempts=199)
    ref = o.forward(x, p, noise)
    assert ref["rc"] == 0
    node = NsdeNode(_cfg(drift, diff, B, max_attempts=199))
    got = node.forward(x, p, noise, keep_tape=True)
    print(f"5120 columns: attempts {got['nattempts']}, draws {got['ndraws']}")
    assert got["nattempts"] == ref["nattempts"] and np.array_equal(got["steps"][:, 3], ref["steps"][:, 3]) and got["ndraws"] == ref["ndraws"]
    assert got["nfe1"] == ref["nfe1"] and got["nfe2"] == ref["nfe2"]
    assert _rel(got["u"], ref["u"]) <= 2e-4 and np.allclose(got["saveval"], ref["saveval"], rtol=5e-4, atol=1e-7)
    rng = np.random.default_rng(2)
    ubar = (rng.standard_normal(x.shape) / B).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 10.0 / len(got["saveval"]), np.float32)
    xb, pb = node.backward(ubar, svbar)
    gr = o.backward(ubar, svbar)
    ex, ep = _rel(xb, gr[0]), _rel(pb, gr[1])
    print(f"5120 columns reverse vs fp32 oracle: x_bar {ex:.2e}, p_bar {ep:.2e}")
    assert ex <= 2e-3 and ep <= 2e-3
    node.close()


# ---- the stiffness-estimate regulariser of the SDE layer (experiments/mnist_nsde.jl:51-61; configs/mnist_nsde.yml:6 ships `type: stiff_est`) -------

@pytest.mark.parametrize("kind,B,tol,ctrl,scale,dscale,replay", [("nsde", 64, 0.14, {}, 2.0, 0.5, False), ("nsde", 37, 0.05, AGGR, 3.0, 1.5, True),
                                                                 ("small", 7, 0.05, AGGR, 2.0, 0.5, False), ("deep", 21, 0.08, AGGR, 2.5, 0.8, True)])
def test_stiffness_estimate_solve_and_reverse_match_oracle(kind, B, tol, ctrl, scale, dscale, replay):
    """RNDE_REG_STIFF on SOSRI2 (= AutoSOSRI2(SOSRI2())): per accepted step the callback records |eigen_est| / 10.6 with
    eigen_est = rms(k4 - k3) / rms(H0_4 - H0_3); once at initialisation 1 / 10.6.  Same noise, same accept / reject sequence as the fp32 oracle;
    saved values to 5e-4 (the quotient of two fp32 norms of differences); gradients for cotangents on u(t1) AND on every saved value against the
    fp64 oracle along the same sequence, within 2e-3 of the largest entry plus the case's own fp32-vs-fp64 spread."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    drift, diff, p, x, noise = _setup(kind, B, 13, 400, scale, dscale)
    o32 = SdeOracle(drift, diff, np.float32, tol, tol, tableau="SOSRI2", reg_kind=2, max_attempts=399, **ctrl)
    r32 = o32.forward(x, p, noise)
    assert r32["rc"] == 0
    node = NsdeNode(_cfg(drift, diff, B, reltol=tol, abstol=tol, solver="SOSRI2", regularize=2, max_attempts=399, **ctrl))
    rp = np.stack([r32["steps"][:, 1], r32["steps"][:, 3]], 1) if replay else None
    got = node.forward(x, p, noise, keep_tape=True, replay=rp)
    if ctrl:
        assert (r32["steps"][:, 3] == 0).sum() >= 3, "this case is meant to exercise rejections"
    assert got["nattempts"] == r32["nattempts"] and np.array_equal(got["steps"][:, 3], r32["steps"][:, 3]) and got["ndraws"] == r32["ndraws"]
    assert len(got["saveval"]) == len(r32["saveval"]) == int(r32["steps"][:, 3].sum()) + 1
    assert got["saveval"][0] == np.float32(1.0) / np.float32(10.6)
    assert (got["saveval"][1:] > 0).all() and np.allclose(got["saveval"], r32["saveval"], rtol=5e-4, atol=1e-7)
    nn = o32.eigen_norms()
    assert np.allclose(got["saveval"][1:], (nn[:, 0] / nn[:, 1]) / 10.6, rtol=5e-4)
    assert _rel(got["u"], r32["u"]) <= 2e-4
    rng = np.random.default_rng(1)
    ubar = rng.standard_normal(x.shape).astype(np.float32) / B
    svbar = (rng.standard_normal(len(r32["saveval"])) * 3.0).astype(np.float32)      # (the saved values are O(0.1): weight them so that their term is a visible share)
    g32 = o32.backward(ubar, svbar)
    o64 = SdeOracle(drift, diff, np.float64, tol, tol, tableau="SOSRI2", reg_kind=2, max_attempts=399, **ctrl)
    o64.set_replay(r32["steps"][:, 1], r32["steps"][:, 3].astype(np.int32))
    assert o64.forward(x, p, noise)["nattempts"] == r32["nattempts"]
    g64 = o64.backward(ubar, svbar)
    g64_u = o64.backward(ubar, np.zeros_like(svbar))
    xb, pb = node.backward(ubar, svbar)
    shares = []
    for name, a, b32, b64, b0 in (("x_bar", xb, g32[0], g64[0], g64_u[0]), ("p_bar", pb, g32[1], g64[1], g64_u[1])):
        sc = np.abs(b64).max()
        spread = np.abs(np.asarray(b32, np.float64) - b64).max()
        err = np.abs(np.asarray(a, np.float64) - b64).max()
        share = np.abs(b64 - b0).max() / sc
        print(f"stiff {kind} B={B} {name}: device-fp64 {err / sc:.2e}, fp32 oracle-fp64 {spread / sc:.2e}, the callback's share of the gradient {share:.2f}")
        shares.append(share)
        assert err <= 2e-3 * sc + 3 * spread, (name, err, sc, spread)
    assert max(shares) > 0.05, "the saved values' cotangent must matter in what is being compared"
    node.close()


def test_stiffness_estimate_config5_full_size_and_the_constant():
    """The shipped NSDE configuration at full size: B = 512, tol 0.14, SOSRI2, `type: stiff_est` with lambda = 0.1 on mean(saveval)
    (mnist_nsde.jl:51-61, :99) -- forward + reverse against the fp64 oracle on an explicit noise pool; and stability_size as a configuration
    value (2.0 instead of 10.6 scales every saved value and nothing else)."""
    from oracle.oracle_sde import SdeOracle
    from tests.util import NsdeNode
    B = 512
    drift, diff, p, x, noise = _setup("nsde", B, 21, 257, scale=1.0, dscale=1.0)
    o64 = SdeOracle(drift, diff, np.float64, tableau="SOSRI2", reg_kind=2, max_attempts=256)
    r64 = o64.forward(x, p, noise)
    assert r64["rc"] == 0
    node = NsdeNode(_cfg(drift, diff, B, solver="SOSRI2", regularize=2, max_attempts=256))
    got = node.forward(x, p, noise, keep_tape=True)
    print(f"config 5 stiff_est: attempts {got['nattempts']} (fp64 oracle {r64['nattempts']}), saved {got['saveval']}")
    assert got["nattempts"] == r64["nattempts"] and np.array_equal(got["steps"][:, 3], r64["steps"][:, 3]) and got["ndraws"] == r64["ndraws"]
    assert _rel(got["u"], r64["u"]) <= 2e-4
    assert np.allclose(got["saveval"], r64["saveval"], rtol=5e-4, atol=1e-7)
    rng = np.random.default_rng(2)
    ubar = (rng.standard_normal(x.shape) / B).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 0.1 / len(got["saveval"]), np.float32)
    xb, pb = node.backward(ubar, svbar)
    g64 = o64.backward(ubar, svbar)
    ex, ep = _rel(xb, g64[0]), _rel(pb, g64[1])
    print(f"config 5 stiff_est reverse vs fp64 oracle: x_bar {ex:.2e}, p_bar {ep:.2e}")
    assert ex <= 1e-3 and ep <= 1e-3
    node2 = NsdeNode(_cfg(drift, diff, B, solver="SOSRI2", regularize=2, max_attempts=256, stability_size=2.0))
    got2 = node2.forward(x, p, noise)
    assert np.array_equal(got2["u"], got["u"]) and np.allclose(got2["saveval"] * 2.0, got["saveval"] * 10.6, rtol=1e-6)
    node.close(); node2.close()


def test_stiffness_estimate_layer_contract():
    """The layer as the unchanged experiment calls it: `func = save_func` (a closure) on a layer built with AutoSOSRI2(SOSRI2()); the closure is
    recognised (mnist_nsde.jl:53-58), the handle records the stiffness estimate, autograd carries lambda * mean(saveval) back.  Refusals: the
    estimate under SOSRI (create-time, C ABI), under a plain SOSRI2 from a closure (the reference would record zeros), an unknown closure."""
    import torch
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    from tests.util import NsdeNode
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    mk = lambda solver: rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g), [0.0, 1.0], True,
                                             solver, save_everystep=False, reltol=1.4e-1, abstol=1.4e-1, save_start=False, max_batch=64)
    nsde = mk("AutoSOSRI2")
    stab = 1.0 / 10.6

    def save_func(u, t, integrator):                     # mnist_nsde.jl:53-58
        s = abs(integrator.eigen_est)
        return stab * (0 if (s == 0 or s != s) else s)
    x = torch.randn(48, 32, generator=g).to(dev)
    p = nsde.p.to(dev).requires_grad_(True)
    nsde.seed = 100
    u, n1, n2, sv = nsde(x, p, func=save_func)
    nsde.seed = 100
    u_e, _, _, sv_e = nsde(x, p, func=lambda u, t, integ: integ.EEst * integ.dt)
    assert torch.equal(u.detach(), u_e.detach()) and n1 == n2 and len(sv.saveval) == len(sv_e.saveval)      # same solve, another recorded value
    assert float(sv.saveval[0].detach()) == np.float32(1.0) / np.float32(10.6) and float(sv_e.saveval[0].detach()) == 0.0
    assert not torch.allclose(sv.saveval[1:], sv_e.saveval[1:])
    loss = u.square().mean() + 0.1 * sv.saveval.mean()
    loss.backward()
    gp = p.grad.clone()
    p.grad = None
    nsde.seed = 100
    u2, _, _, sv2 = nsde(x, p, func="stiff_est")
    (u2.square().mean() + 0.1 * sv2.saveval.mean()).backward()
    assert torch.equal(gp, p.grad) and torch.isfinite(gp).all() and float(gp.abs().max()) > 0
    p.grad = None
    nsde.seed = 100
    u3, _, _, sv3 = nsde(x, p, func="stiff_est")
    u3.square().mean().backward()
    assert not torch.equal(gp, p.grad)                   # the callback's term is in the gradient
    with pytest.raises(ValueError, match="composite"):
        mk("SOSRI2")(x, p, func=save_func)
    with pytest.raises(ValueError, match="SOSRI2"):
        mk("SOSRI")(x, p, func="stiff_est")
    with pytest.raises(ValueError, match="composite"):      # the NAME takes the same check as the closure: plain SOSRI2 leaves eigen_est at its initial value
        mk("SOSRI2")(x, p, func="stiff_est")
    with pytest.raises(ValueError, match="none of the callbacks"):
        nsde(x, p, func=lambda u, t, integ: integ.dt)
    drift, diff, _, _, _ = _setup("nsde", 16, 1, 1)
    with pytest.raises(_lib.RndeError, match="SOSRI2 only"):
        NsdeNode(_cfg(drift, diff, 16, solver="SOSRI", regularize=2))
