"""GPU: the host-side mirror (TrackedNeuralODE / ClassifierNODE) end to end through torch.autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(rn, D=36, Hd=10, B=12, tol=1e-3, regularize=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    dyn = rn.MLPDynamics(D, Hd, generator=g)
    for l in dyn.layers:
        l.W.mul_(3.0)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, regularize, "Tsit5", save_everystep=False, reltol=tol, abstol=tol,
                               save_start=False, max_batch=B, max_attempts=64)
    return node, g


def test_layer_call_contract_and_autograd_vs_oracle(rnde):
    from oracle.oracle import Oracle, arch_mnist
    rn = rnde
    D, Hd, B = 36, 10, 12
    node, g = _model(rn, D, Hd, B)
    x = torch.rand(B, D, generator=g).cuda().requires_grad_(True)
    p = node.p.cuda().clone().requires_grad_(True)
    u, nfe, sv = node(x, p)                                     # (res, nfe, sv), neural_ode.jl:143
    assert u.shape == (B, D) and isinstance(nfe, int) and nfe % 6 == 3 and sv.saveval.ndim == 1
    w = torch.randn(B, D, generator=g).cuda()
    loss = (u * w).sum() + 40.0 * sv.saveval.sum()              # test/test_node.jl:47-57 style objective
    loss.backward()
    orc = Oracle(arch_mnist(D, Hd), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r = orc.forward(x.detach().cpu().numpy().astype(np.float64), p.detach().cpu().numpy().astype(np.float64))
    assert r["nfe"] == nfe
    xb, pb, _ = orc.backward(w.cpu().numpy().astype(np.float64), np.full(len(r["saveval"]), 40.0))
    assert np.abs(u.detach().cpu().numpy() - r["u"]).max() < 2e-5
    assert np.abs(x.grad.cpu().numpy() - xb).max() <= 3e-3 * np.abs(xb).max()
    assert np.abs(p.grad.cpu().numpy() - pb).max() <= 3e-3 * np.abs(pb).max()


def test_saveat_layer_returns_3d_array_and_differentiates(rnde):
    """{true,true} method (neural_ode.jl:146-180) + per-call saveat override (update_saveat!, :35-46)."""
    from oracle.oracle import Oracle, arch_mnist
    rn = rnde
    D, Hd, B = 36, 10, 12
    g = torch.Generator().manual_seed(1)
    dyn = rn.MLPDynamics(D, Hd, generator=g)
    for l in dyn.layers:
        l.W.mul_(3.0)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", reltol=1e-3, abstol=1e-3, saveat=[0.0, 0.5, 1.0],
                               max_batch=B, max_attempts=64)
    x = torch.rand(B, D, generator=g).cuda().requires_grad_(True)
    p = node.p.cuda().clone().requires_grad_(True)
    u0, _, _ = node(x, p)
    assert u0.shape == (B, 3, D)
    assert torch.equal(u0[:, 0], x.detach())                   # save point at t0 is the input itself
    sa = [0.1, 0.35, 0.62, 0.97]
    u, nfe, sv = node(x, p, saveat=sa)
    assert u.shape == (B, 4, D) and node.kwargs["saveat"] == [0.0, 0.5, 1.0]   # restored (neural_ode.jl:41-44)
    w = torch.randn(B, 4, D, generator=g).cuda()
    ((u * w).sum() + 40.0 * sv.saveval.sum()).backward()
    orc = Oracle(arch_mnist(D, Hd), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r = orc.forward(x.detach().cpu().numpy().astype(np.float64), p.detach().cpu().numpy().astype(np.float64),
                    saveat=np.array(sa, dtype=np.float32))
    assert r["nfe"] == nfe
    xb, pb, _ = orc.backward(w.cpu().numpy().astype(np.float64), np.full(len(r["saveval"]), 40.0))
    assert np.abs(u.detach().cpu().numpy() - r["u"]).max() < 3e-5
    assert np.abs(x.grad.cpu().numpy() - xb).max() <= 3e-3 * np.abs(xb).max()
    assert np.abs(p.grad.cpu().numpy() - pb).max() <= 3e-3 * np.abs(pb).max()


def test_save_everystep_layer_returns_every_accepted_state(rnde):
    """The other way to `return_multiple` (neural_ode.jl:10-11: save_everystep = true): rnde_node_forward_everystep returns the state after every
    accepted step (the initial one first, save_start = true by default): the count comes back with the call, the last state is the plain solve's
    end state bit for bit, and values and gradients are those of the saveat call at the same times (the saved value at a step's end is u_new)."""
    from oracle.oracle import Oracle, arch_mnist
    rn = rnde
    D, Hd, B = 36, 10, 12
    g = torch.Generator().manual_seed(3)
    dyn = rn.MLPDynamics(D, Hd, generator=g)
    for l in dyn.layers:
        l.W.mul_(3.0)
    kw = dict(reltol=1e-3, abstol=1e-3, max_batch=B, max_attempts=64)
    every = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=True, **kw)
    plain = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, **kw)
    x = torch.rand(B, D, generator=g).cuda().requires_grad_(True)
    p = every.p.cuda().clone().requires_grad_(True)
    u, nfe, sv = every(x, p)
    ts = every.last_times
    n = len(ts)
    assert u.shape == (B, n, D) and n >= 4 and ts[0] == 0.0 and ts[-1] == 1.0 and all(b > a for a, b in zip(ts, ts[1:]))
    assert torch.equal(u[:, 0], x.detach())
    with torch.no_grad():
        ue, nfe_e, _ = plain(x, p)
    assert nfe == nfe_e and torch.equal(u[:, -1].detach(), ue) and (nfe - 3) // 6 >= n - 1      # one saved state per accepted step (+ the start)
    w = torch.randn(B, n, D, generator=g).cuda()
    ((u * w).sum() + 40.0 * sv.saveval.sum()).backward()
    gx, gp = x.grad.clone(), p.grad.clone()
    x.grad = None; p.grad = None
    at = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", saveat=ts, **kw)
    u2, nfe2, sv2 = at(x, p)
    assert nfe2 == nfe and torch.equal(u2.detach(), u.detach())
    ((u2 * w).sum() + 40.0 * sv2.saveval.sum()).backward()
    assert torch.equal(x.grad, gx) and torch.equal(p.grad, gp)
    orc = Oracle(arch_mnist(D, Hd), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r = orc.forward(x.detach().cpu().numpy().astype(np.float64), p.detach().cpu().numpy().astype(np.float64), saveat=np.array(ts, dtype=np.float32))
    assert r["nfe"] == nfe and np.abs(u.detach().cpu().numpy() - r["u"]).max() < 3e-5
    # a result that does not fit the room given is refused with the room needed
    import ctypes as C
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    h = every._acquire(x.detach(), False)
    small = torch.empty(B * 2 * D, device="cuda")
    nout = C.c_int32(0)
    st = L.rnde_node_forward_everystep(h.ptr, x.detach().data_ptr(), p.detach().data_ptr(), B, 0.0, 1.0, 1, small.data_ptr(), 2, None, C.byref(nout), None, None, None, 0, None)
    assert st != 0 and nout.value == n


def test_latent_ode_layer_call(rnde):
    """The node of experiments/latent_ode.jl:113-147: gen_dynamics (tanh + 8 Dense), time independent, saveat = the data's
    time grid; per-call saveat override as in loss_function (latent_ode.jl:237-241).  Runs on the chain engine."""
    from oracle.oracle import Oracle, arch_latent
    rn = rnde
    B = 24
    g = torch.Generator().manual_seed(2)
    dyn = rn.LatentGenDynamics(generator=g)
    for l in dyn.layers:
        l.W.mul_(1.5)            # truncation-dominated regime (with Glorot weights fp32 EEst sits on its rounding floor)
    grid = np.linspace(0.0, 1.0, 49).astype(np.float32)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], False, True, "Tsit5", saveat=grid.tolist(), reltol=1e-3, abstol=1e-3,
                               max_batch=B, max_attempts=64)
    assert node.P == 8280 and node.return_multiple
    z0 = (torch.rand(B, 20, generator=g) * 2 - 1).cuda().requires_grad_(True)
    p = node.p.cuda().clone().requires_grad_(True)
    res, nfe, sv = node(z0, p)
    assert res.shape == (B, 49, 20) and sv.saveval.ndim == 1
    sub = grid[::6]
    res2, nfe2, sv2 = node(z0, p, saveat=sub.tolist())
    assert nfe2 == nfe and torch.allclose(res2, res[:, ::6], atol=1e-6)
    w = torch.randn(B, 49, 20, generator=g).cuda()
    ((res * w).sum() + 5.0 * sv.saveval.sum()).backward()
    orc = Oracle(arch_latent(), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r = orc.forward(z0.detach().cpu().numpy().astype(np.float64), p.detach().cpu().numpy().astype(np.float64), saveat=grid)
    assert r["nfe"] == nfe
    xb, pb, _ = orc.backward(w.cpu().numpy().astype(np.float64), np.full(len(r["saveval"]), 5.0))
    # conditioning: spread between the fp32 and fp64 oracles on the same inputs (8 stacked tanh layers amplify rounding)
    o32 = Oracle(arch_latent(), np.float32, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r32 = o32.forward(z0.detach().cpu().numpy(), p.detach().cpu().numpy(), saveat=grid)
    assert r32["nfe"] == nfe
    xb32, pb32, _ = o32.backward(w.cpu().numpy(), np.full(len(r["saveval"]), 5.0, dtype=np.float32))
    su = np.abs(r32["u"] - r["u"]).max()
    sx, sp = np.abs(xb32 - xb).max() / np.abs(xb).max(), np.abs(pb32 - pb).max() / np.abs(pb).max()
    assert np.abs(res.detach().cpu().numpy() - r["u"]).max() < 3e-5 + 4 * su
    assert np.abs(z0.grad.cpu().numpy() - xb).max() <= (3e-3 + 4 * sx) * np.abs(xb).max()
    assert np.abs(p.grad.cpu().numpy() - pb).max() <= (3e-3 + 4 * sp) * np.abs(pb).max()


def test_unregularised_layer_returns_nothing_for_sv(rnde):
    node, g = _model(rnde, regularize=False)
    x = torch.rand(5, 36, generator=g).cuda()
    with torch.no_grad():
        u, nfe, sv = node(x)                                     # {false,false} method, neural_ode.jl:48-77
    assert sv is None and nfe % 6 == 3 and torch.isfinite(u).all()


def test_nfe_probe_between_forward_and_backward_keeps_the_tape(rnde):
    """experiments/mnist_node.jl:245 probes NFE with a plain call; it must not clobber a pending backward."""
    node, g = _model(rnde)
    x = torch.rand(8, 36, generator=g).cuda()
    p = node.p.cuda().clone().requires_grad_(True)
    u, nfe, sv = node(x, p)
    with torch.no_grad():
        node(torch.rand(8, 36, generator=g).cuda(), p)
    (u.sum() + sv.saveval.mean()).backward()
    assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0


def test_classifier_training_reduces_loss(rnde):
    rn = rnde
    g = torch.Generator().manual_seed(3)
    dyn = rn.MLPDynamics(784, 100, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1e-3, abstol=1e-3,
                               save_start=False, max_batch=32, max_attempts=64)
    model = rn.ClassifierNODE(node, rn.Dense(784, 10, generator=g))
    opt = rn.FluxOptimiser(model.trainable())
    x = torch.rand(32, 1, 28, 28, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (32,), generator=g)].cuda()
    losses = []
    for _ in range(12):
        loss, ce, reg, nfe = rn.loss_function(x, y, model, lam=10.0)
        loss.backward()
        opt.step()
        losses.append(float(ce))
    assert losses[-1] < 0.7 * losses[0], losses
    assert rn.accuracy(model, [(x, y)]) > 0.3


def test_fused_head_step_matches_autograd(rnde):
    """rnde_classifier_head + fused_loss_and_grad against the torch.autograd path on the same inputs."""
    rn = rnde
    def make():
        g = torch.Generator().manual_seed(5)
        dyn = rn.MLPDynamics(784, 100, generator=g)
        node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1e-3, abstol=1e-3,
                                   save_start=False, max_batch=24, max_attempts=64)
        post = rn.Dense(784, 10, generator=g)
        post.b.uniform_(-0.1, 0.1, generator=g)
        model = rn.ClassifierNODE(node, post)
        x = torch.rand(24, 1, 28, 28, generator=g).cuda()
        y = torch.eye(10)[torch.randint(0, 10, (24,), generator=g)].cuda()
        return model, x, y
    m1, x, y = make()
    loss1, ce1, reg1, nfe1 = rn.loss_function(x, y, m1, lam=50.0)
    loss1.backward()
    m2, _, _ = make()
    loss2, ce2, reg2, nfe2 = rn.fused_loss_and_grad(m2, x, y, lam=50.0)
    assert nfe1 == nfe2
    assert abs(float(loss1.detach()) - loss2) <= 1e-5 * max(1.0, abs(loss2))
    for a, b in ((m1.p2.grad, m2.p2.grad), (m1.p3.grad, m2.p3.grad)):
        assert (a - b).abs().max() <= 5e-4 * b.abs().max()   # different fp32 association in the head GEMM


@pytest.mark.parametrize("reg,sync", [("error_est", True), ("stiff_est", True), ("stiff_est", False), ("error_stiff_est", False), ("closure", True)])
def test_fused_step_serves_every_regulariser_of_the_experiment(rnde, reg, sync):
    """`fused_loss_and_grad(func=, agg=)` against `loss_function` + torch.autograd for each `type` of experiments/mnist_node.jl:62-103: error_est (mean),
    stiff_est (`maximum`, AutoTsit5(Tsit5())), error_stiff_est (mean), and the stiffness callback passed as the script passes it -- a closure.
    Same loss, same regulariser value, gradients to 5e-4 of the largest entry (the head GEMM associates differently)."""
    rn = rnde
    solver = "Tsit5" if reg == "error_est" else "AutoTsit5"

    def make():
        g = torch.Generator().manual_seed(5)
        dyn = rn.MLPDynamics(784, 100, generator=g)
        node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, solver, save_everystep=False, reltol=1e-3, abstol=1e-3,
                                   save_start=False, max_batch=24, max_attempts=64)
        post = rn.Dense(784, 10, generator=g)
        post.b.uniform_(-0.1, 0.1, generator=g)
        model = rn.ClassifierNODE(node, post)
        x = torch.rand(24, 1, 28, 28, generator=g).cuda()
        y = torch.eye(10)[torch.randint(0, 10, (24,), generator=g)].cuda()
        return model, x, y
    stab = 1.0 / 3.5068

    def save_func(u, t, integrator):                      # mnist_node.jl:74-79
        s_ = abs(integrator.eigen_est)
        return stab * (0 if (s_ == 0 or s_ != s_) else s_)
    if reg == "closure":
        func, agg_t, agg_s, lam = save_func, torch.max, "max", 0.1
    else:
        lam0, lam1, func, agg_t, _ = rn.REGULARISERS[reg]
        agg_s, lam = ("max" if agg_t is torch.max else "mean"), lam0
    m1, x, y = make()
    loss1, ce1, reg1, nfe1 = rn.loss_function(x, y, m1, lam=lam, agg=agg_t, func=func)
    loss1.backward()
    m2, _, _ = make()
    loss2, ce2, reg2, nfe2 = rn.fused_loss_and_grad(m2, x, y, lam=lam, func=func, agg=agg_s, sync=sync)
    torch.cuda.synchronize()
    assert nfe1 == nfe2
    assert abs(float(reg1) - float(reg2)) <= 1e-5 * max(1e-3, abs(float(reg1))) and float(reg2) > 0
    assert abs(float(loss1.detach()) - float(loss2)) <= 1e-5 * max(1.0, abs(float(loss2)))
    for a, b in ((m1.p2.grad, m2.p2.grad), (m1.p3.grad, m2.p3.grad)):
        assert (a - b).abs().max() <= 5e-4 * b.abs().max()


def test_async_backward_matches_the_synchronous_one(rnde):
    """rnde_node_backward_async (fused step with sync=False): same gradients as the synchronising call, valid in stream order."""
    rn = rnde
    g = torch.Generator().manual_seed(5)
    dyn = rn.MLPDynamics(36, 10, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1e-3, abstol=1e-3, save_start=False,
                               max_batch=16, max_attempts=64)
    model = rn.ClassifierNODE(node, rn.Dense(36, 10, "identity", generator=g), device=torch.device("cuda", 0))
    x = torch.rand(16, 36, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (16,), generator=g)].cuda()
    l1, ce1, reg1, nfe1 = rn.fused_loss_and_grad(model, x, y, lam=50.0)
    g2, g3 = model.p2.grad.clone(), model.p3.grad.clone()
    l2, ce2, reg2, nfe2 = rn.fused_loss_and_grad(model, x, y, lam=50.0, sync=False)
    assert torch.is_tensor(l2) and nfe1 == nfe2
    torch.cuda.synchronize()
    assert torch.equal(model.p2.grad, g2) and torch.equal(model.p3.grad, g3)
    assert abs(float(l2) - l1) <= 1e-6 * max(1.0, abs(l1))


def test_one_call_step_gradient_equals_the_three_calls_at_the_headline_shape(rnde):
    """rnde_node_classifier_grad (forward + head + reverse in one call, head and reverse packs queued before the forward's host wait)
    against rnde_node_forward + rnde_classifier_head + rnde_node_backward: the same kernels on the same inputs in the same order --
    loss, NFE and both gradients bit for bit, MNIST shape, B = 512, tol 1.4e-8; twice, so a stale pack or tape would show."""
    rn = rnde
    g = torch.Generator().manual_seed(11)
    dyn = rn.MLPDynamics(784, 100, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", reltol=1.4e-8, abstol=1.4e-8, max_batch=512, max_attempts=256)
    model = rn.ClassifierNODE(node, rn.Dense(784, 10, generator=g), device=torch.device("cuda", 0))
    x = torch.rand(512, 1, 28, 28, generator=g).cuda()
    y = torch.eye(10)[torch.randint(0, 10, (512,), generator=g)].cuda()
    for rep in range(2):
        l1, ce1, reg1, nfe1 = rn.fused_loss_and_grad(model, x, y, lam=100.0)               # three calls, synchronising
        g2, g3 = model.p2.grad.clone(), model.p3.grad.clone()
        l2, ce2, reg2, nfe2 = rn.fused_loss_and_grad(model, x, y, lam=100.0, sync=False)   # one call
        torch.cuda.synchronize()
        assert nfe1 == nfe2 and reg1 == pytest.approx(reg2, rel=1e-6) and float(ce2) == ce1
        assert torch.equal(model.p2.grad, g2) and torch.equal(model.p3.grad, g3)
        with torch.no_grad():
            model.p2.add_(0.01 * torch.randn(model.p2.shape, generator=g).cuda())           # new weights: the packs must follow
    for hs in node._handles.values():
        for h in hs:
            assert rn._lib.lib().rnde_node_fallback_count(h.ptr) == 0


@pytest.mark.parametrize("B,ncls,reg", [(37, 7, True), (5, 10, False), (48, 3, True)])
def test_one_call_step_gradient_ragged_and_unregularised(rnde, B, ncls, reg):
    """rnde_node_classifier_grad off the headline shape: ragged batches (the padded columns must stay out of every sum), other class
    counts, lambda = 0 (no regulariser cotangent), generic-geometry kernels (D = 36, H = 10) -- against the three separate calls."""
    rn = rnde
    g = torch.Generator().manual_seed(100 + B)
    dyn = rn.MLPDynamics(36, 10, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1e-4, abstol=1e-4, save_start=False,
                               max_batch=64, max_attempts=64)
    model = rn.ClassifierNODE(node, rn.Dense(36, ncls, "identity", generator=g), device=torch.device("cuda", 0))
    x = torch.rand(B, 36, generator=g).cuda()
    y = torch.eye(ncls)[torch.randint(0, ncls, (B,), generator=g)].cuda()
    l1, ce1, reg1, nfe1 = rn.fused_loss_and_grad(model, x, y, lam=25.0, regularize=reg)
    g2, g3 = model.p2.grad.clone(), model.p3.grad.clone()
    l2, ce2, reg2, nfe2 = rn.fused_loss_and_grad(model, x, y, lam=25.0, regularize=reg, sync=False)
    torch.cuda.synchronize()
    assert nfe1 == nfe2 and float(ce2) == ce1 and reg1 == pytest.approx(reg2, rel=1e-6) and (reg or reg2 == 0.0)
    assert torch.equal(model.p2.grad, g2) and torch.equal(model.p3.grad, g3)
    assert torch.isfinite(model.p2.grad).all() and float(model.p2.grad.abs().max()) > 0


def _latent_batch(g, B, T=49, in_dim=37):
    data = torch.randn(B, T, in_dim, generator=g)
    mask = (torch.rand(B, T, in_dim, generator=g) > 0.7).float()
    mask[:, 0, 0] = 1
    grid = torch.linspace(0, 1, T)
    t = grid.reshape(1, T, 1).repeat(B, 1, 1)
    return data.cuda(), mask.cuda(), t.cuda(), grid


def test_latent_time_series_model_end_to_end(rnde):
    """LatentTimeSeriesModel (time_series.jl:40-70) at the reference's sizes around the device solve: shapes of the five results,
    and the node's part of the total-loss gradient (z0-bar and p3-bar given the cotangents the decoder/likelihood/regulariser hand
    it) against the fp64 oracle run on the same z0 -- i.e. the hot path checked inside its real caller."""
    from oracle.oracle import Oracle, arch_latent
    rn = rnde
    g = torch.Generator().manual_seed(12)
    B = 16
    data, mask, t, grid = _latent_batch(g, B)
    model = rn.build_latent_ode(saveat=grid.tolist(), generator=g, reltol=1e-3, abstol=1e-3, max_batch=B, max_attempts=64)
    with torch.no_grad():
        model.p3.mul_(1.5)                          # truncation-dominated regime, as in test_latent_ode_layer_call
    assert [p.numel() for p in model.trainable()] == [3 * (175 * 40 + 40) + 2 * (40 * 50 + 50) + 40 * 100 + 100, 100 * 50 + 50 + 50 * 40 + 40,
                                                       8280, 20 * 37 + 37]
    _, tt, trow = rn.get_t_saveat(t, grid)
    seen = {}
    inner = model.node
    class Spy:                                       # records what the model hands to / gets from the node
        def __call__(self, z0, p3, **kw):
            z0.retain_grad()
            res, nfe, sv = inner(z0, p3, **kw)
            res.retain_grad(); sv.saveval.retain_grad()
            seen.update(z0=z0, res=res, sv=sv.saveval, nfe=nfe)
            return res, nfe, sv
    model.node = Spy()
    gen = torch.Generator(device="cuda").manual_seed(1)
    total, nll, kl, reg, nfe = rn.latent_loss_function(data, mask, trow, model, lam_r=1.0e2, lam_k=0.5, saveat=tt, generator=gen)
    assert torch.isfinite(total) and nfe % 6 == 3 and seen["res"].shape == (B, 49, 20)
    total.backward()
    for p in model.trainable():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    z0 = seen["z0"].detach().cpu().numpy().astype(np.float64)
    p3 = model.p3.detach().cpu().numpy().astype(np.float64)
    orc = Oracle(arch_latent(), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r = orc.forward(z0, p3, saveat=grid.numpy())
    assert r["nfe"] == nfe
    xb, pb, _ = orc.backward(seen["res"].grad.cpu().numpy().astype(np.float64), seen["sv"].grad.cpu().numpy().astype(np.float64))
    o32 = Oracle(arch_latent(), np.float32, reltol=1e-3, abstol=1e-3, reg_kind=1)
    r32 = o32.forward(z0.astype(np.float32), p3.astype(np.float32), saveat=grid.numpy())
    assert r32["nfe"] == nfe
    xb32, pb32, _ = o32.backward(seen["res"].grad.cpu().numpy(), seen["sv"].grad.cpu().numpy())
    sx, sp = np.abs(xb32 - xb).max() / np.abs(xb).max(), np.abs(pb32 - pb).max() / np.abs(pb).max()
    assert np.abs(seen["res"].detach().cpu().numpy() - r["u"]).max() < 3e-5 + 4 * np.abs(r32["u"] - r["u"]).max()
    assert np.abs(seen["z0"].grad.cpu().numpy() - xb).max() <= (3e-3 + 4 * sx) * np.abs(xb).max()
    assert np.abs(model.p3.grad.cpu().numpy() - pb).max() <= (3e-3 + 4 * sp) * np.abs(pb).max()


def test_latent_ode_training_reduces_loss(rnde):
    """A few AdaMax steps of experiments/latent_ode.jl's training loop (loss_function + Optimiser(InvDecay, AdaMax)) on one
    synthetic batch: the negative log likelihood must fall."""
    rn = rnde
    g = torch.Generator().manual_seed(13)
    B = 16
    data, mask, t, grid = _latent_batch(g, B)
    data = 0.1 * data
    model = rn.build_latent_ode(saveat=grid.tolist(), generator=g, reltol=1e-3, abstol=1e-3, max_batch=B, max_attempts=96)
    opt = rn.FluxAdaMax(model.trainable())
    _, tt, trow = rn.get_t_saveat(t, grid)
    gen = torch.Generator(device="cuda").manual_seed(2)
    nlls = []
    for it in range(10):
        total, nll, kl, reg, nfe = rn.latent_loss_function(data, mask, trow, model, lam_r=1.0e2, lam_k=rn.lambda_k(it), saveat=tt,
                                                           generator=gen)
        total.backward()
        opt.step()
        nlls.append(float(nll))
    assert nlls[-1] < 0.8 * nlls[0], nlls


def test_momentum_step_matches_the_torch_recurrence(rnde):
    """rnde_momentum_step (one launch per parameter group) against the host form of Optimiser(InvDecay, Momentum),
    reference src/utils.jl:149-156 + experiments/mnist_node.jl:130."""
    rn = rnde
    g = torch.Generator().manual_seed(21)
    p0 = torch.randn(166_418 - 7850, generator=g)
    pc, pg = p0.clone().requires_grad_(True), p0.clone().cuda().requires_grad_(True)
    oc, og = rn.FluxOptimiser([torch.zeros(0), pc]), rn.FluxOptimiser([torch.zeros(0, device="cuda"), pg])
    for k in range(5):
        grad = torch.randn(p0.shape, generator=g)
        pc.grad, pg.grad = grad.clone(), grad.clone().cuda()
        oc.step(); og.step()
        assert pg.grad is None and og.n == oc.n
    assert (pg.detach().cpu() - pc.detach()).abs().max() <= 1e-6 * pc.detach().abs().max()
    assert (og.v[0].cpu() - oc.v[0]).abs().max() <= 1e-6 * oc.v[0].abs().max()


def test_layer_with_the_dp5_pair(rnde):
    """solver="DP5" on the host mirror: the small MNIST-form network routed to the tableau-as-data kernels (col_tile 65)."""
    from oracle.oracle import Oracle, arch_mnist
    rn = rnde
    D, Hd, B = 36, 10, 12
    g = torch.Generator().manual_seed(3)
    dyn = rn.MLPDynamics(D, Hd, generator=g)
    for l in dyn.layers:
        l.W.mul_(3.0)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "DP5", save_everystep=False, reltol=1e-3, abstol=1e-3, save_start=False,
                               max_batch=B, max_attempts=64, col_tile=65)
    x = torch.rand(B, D, generator=g).cuda().requires_grad_(True)
    p = node.p.cuda().clone().requires_grad_(True)
    u, nfe, sv = node(x, p)
    w = torch.randn(B, D, generator=g).cuda()
    ((u * w).sum() + 40.0 * sv.saveval.sum()).backward()
    orc = Oracle(arch_mnist(D, Hd), np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1, solver="DP5")
    r = orc.forward(x.detach().cpu().numpy().astype(np.float64), p.detach().cpu().numpy().astype(np.float64))
    assert r["nfe"] == nfe
    xb, pb, _ = orc.backward(w.cpu().numpy().astype(np.float64), np.full(len(r["saveval"]), 40.0))
    assert np.abs(u.detach().cpu().numpy() - r["u"]).max() < 2e-5
    assert np.abs(x.grad.cpu().numpy() - xb).max() <= 3e-3 * np.abs(xb).max()
    assert np.abs(p.grad.cpu().numpy() - pb).max() <= 3e-3 * np.abs(pb).max()


def test_adam_step_matches_the_torch_recurrence(rnde):
    """rnde_adam_step (Flux.Optimise.ADAM: the optimiser of experiments/mnist_nsde.jl) against torch.optim.Adam, five steps."""
    rn = rnde
    g = torch.Generator().manual_seed(5)
    p1 = torch.randn(5248, generator=g).cuda().requires_grad_(True)
    p2 = p1.detach().clone().requires_grad_(True)
    opt1 = torch.optim.Adam([p1], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    opt2 = rn.FluxADAM([p2], eta=0.01)
    for k in range(5):
        gr = torch.randn(5248, generator=g).cuda() * (1.0 + k)
        p1.grad = gr.clone(); p2.grad = gr.clone()
        opt1.step(); opt2.step()
    torch.cuda.synchronize()
    assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-6), float((p1 - p2).abs().max())


def test_error_estimate_regulariser_lowers_nfe_at_held_accuracy(monkeypatch):
    """A REGRESSION test of the training loop, not evidence for the method: in matrix mode 0 with seed 1999 the regularised run ends at a lower NFE; over three
    seeds and both matrix modes the sign of that effect is not stable on this synthetic set (profiles/r06_train_synth_seeds.json: at tol 1.4e-8 the
    fp32 error estimate the regulariser trains on is rounding noise).  Pinned to the mode and seed the numbers below were measured in.
    The paper's claim on THIS implementation, shortened (tools/train_synth.py is the full record, profiles/r03_train_synth.json): the
    reference's training loop (experiments/mnist_node.jl:220-263 -- lambda 100 -> 10, InvDecay/Momentum, NFE probe on the fixed first
    batch, accuracy of src/metrics.jl:4-18) on a learnable synthetic 10-class set, 4 epochs of 24 batches of 512 (lambda decays 100 -> 10 over
    these 4 epochs), vanilla against the error-estimate regulariser: the regularised model needs FEWER function evaluations at an accuracy no
    more than 2.5 points lower (measured: NFE 615 -> 531 at 92.2 / 93.4 % test accuracy; the 10-epoch record: 597 -> 525 at 93.5 / 92.6 %)."""
    import importlib.util
    import os
    import torch
    monkeypatch.setenv("RNDE_X3", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("train_synth", os.path.join(root, "tools", "train_synth.py"))
    ts = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(ts)
    dev = torch.device("cuda", 0)
    tr, te, _ = ts.synthetic_set(24 * 512, 8 * 512, 1999)
    train, test = ts.batches_of(*tr, dev), ts.batches_of(*te, dev)
    van = ts.run("vanilla", train, test, 4, dev, 1999, 1000, False)
    err = ts.run("error_est", train, test, 4, dev, 1999, 1000, False)
    assert "failed" not in van and "failed" not in err
    print("vanilla", van["final"], "error_est", err["final"])
    assert err["final"]["nfe"] < van["final"]["nfe"]
    assert err["final"]["test_acc"] >= van["final"]["test_acc"] - 2.5 and err["final"]["test_acc"] > 80.0
