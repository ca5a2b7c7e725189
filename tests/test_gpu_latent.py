"""The latent-ODE caller of the hot path on the device (csrc/rnde_latent.h, include/rnde.h: rnde_latent_*) against the fp64 numpy restatement
(oracle/latent_oracle.py, itself pinned by finite differences in tests/test_host.py) -- NOT against the torch mirror.

Reference: LatentGRU / single_run (experiments/latent_ode.jl:39-106), rec_to_gen + sampling (src/models/time_series.jl:50-59), gen_to_data,
log_likelihood, kl_divergence (latent_ode.jl:148, :192-204, :226-233) and the reverse pass Tracker.gradient performs over them.
fp32 tolerances: forward values 5e-6 of the largest entry (49 recurrent steps of tanh / sigmoid stacks), gradients 1e-5 of the largest
entry of each parameter group and of each GRU layer (sums over 49 x B samples of fp32 products, fixed order); observed 1e-7 .. 6e-7."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(B, T, seed, scale=1.0):
    from oracle import latent_oracle as lo
    rng = np.random.default_rng(seed)
    S = lo.GruShape(37, 40, 50)
    x = np.zeros((B, T, 75))
    x[:, :, :37] = rng.standard_normal((B, T, 37))
    x[:, :, 37:74] = rng.uniform(size=(B, T, 37)) < 0.3
    x[:, :, 74] = np.abs(rng.standard_normal((B, T))) * 0.02
    x[:, T // 3, 37:] = 0.0                                   # a save time with no observation and dt = 0: the state passes through (latent_ode.jl:91-94)
    x[:, 0, 37] = 1.0                                          # (every sample observes something: the likelihood divides by the count)
    def dense(n_in, n_out):
        lim = scale * np.sqrt(6.0 / (n_in + n_out))
        return np.concatenate([rng.uniform(-lim, lim, n_in * n_out), rng.uniform(-0.05, 0.05, n_out)])
    p1 = np.concatenate([dense(175, 40), dense(40, 50), dense(175, 40), dense(40, 50), dense(175, 40), dense(40, 100)])
    p2 = np.concatenate([dense(100, 50), dense(50, 40)])
    p4 = dense(20, 37)
    eps = rng.standard_normal((B, 20))
    res = 0.5 * rng.standard_normal((B, T, 20))              # stands in for the layer call's saved states (the solve itself: tests/test_gpu_chain.py)
    z0b = 0.1 * rng.standard_normal((B, 20))                  # ... and for the cotangent its reverse pass returns
    return S, x, p1, p2, p4, eps, res, z0b


@pytest.mark.parametrize("B,T,seed", [(48, 49, 1), (37, 49, 2), (16, 7, 3), (200, 12, 4), (512, 49, 5)])      # the last: config 4's own size (experiments/latent_ode.jl, batch 512 x 49 times)
def test_latent_caller_matches_the_fp64_restatement(B, T, seed):
    import torch
    from oracle import latent_oracle as lo
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    S, x, p1, p2, p4, eps, res, z0b = _case(B, T, seed)
    lam_k = 0.7
    # ---- oracle (fp64) ----
    y, tape1 = lo.gru_forward(S, p1, x)
    z0, mu0, lv, tape2 = lo.encode_forward(p2, y, eps)
    data, mask = x[:, :, :37], x[:, :, 37:74]
    nll, resb, p4b, ll = lo.decode_loss(p4, res, data, mask)
    kl = lo.kl_per_sample(mu0, lv).mean()
    yb, p2b = lo.encode_backward(p2, tape2, z0b, lam_k / B)
    p1b = lo.gru_backward(S, p1, tape1, yb)
    # ---- device, through the C ABI ----
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    h = C.c_void_p()
    cfg = _lib.LatentConfig(max_batch=B, max_T=T, device=0)
    _lib.check_latent(None, L.rnde_latent_create(C.byref(cfg), C.byref(h)))
    xd, p1d, p2d, p4d, epsd, resd, z0bd = dev(x), dev(p1), dev(p2), dev(p4), dev(eps), dev(res), dev(z0b)
    z0d, mu0d, lvd = torch.empty(B, 20, device="cuda"), torch.empty(B, 20, device="cuda"), torch.empty(B, 20, device="cuda")
    loss2, resbd, p4bd = torch.empty(2, device="cuda"), torch.empty(B, T, 20, device="cuda"), torch.empty(777, device="cuda")
    p1bd, p2bd = torch.empty(29320, device="cuda"), torch.empty(7090, device="cuda")
    torch.cuda.synchronize()
    _lib.check_latent(h, L.rnde_latent_encode(h, xd.data_ptr(), p1d.data_ptr(), p2d.data_ptr(), epsd.data_ptr(), B, T, z0d.data_ptr(), mu0d.data_ptr(), lvd.data_ptr(), None))
    _lib.check_latent(h, L.rnde_latent_decode_loss(h, resd.data_ptr(), p4d.data_ptr(), xd.data_ptr(), B, T, loss2.data_ptr(), resbd.data_ptr(), p4bd.data_ptr(), None))
    _lib.check_latent(h, L.rnde_latent_encode_backward(h, z0bd.data_ptr(), lam_k, p1d.data_ptr(), p2d.data_ptr(), xd.data_ptr(), p1bd.data_ptr(), p2bd.data_ptr(), None))
    torch.cuda.synchronize()
    L.rnde_latent_destroy(h)

    def close(got, ref, tol, what):
        got = got.detach().cpu().numpy().astype(np.float64).reshape(ref.shape)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        print(f"{what}: rel err {err:.2e}")
        assert np.isfinite(got).all() and err <= tol, (what, err)

    close(mu0d, mu0, 5e-6, "mu0"); close(lvd, lv, 5e-6, "logvar"); close(z0d, z0, 5e-6, "z0")
    close(loss2, np.array([nll, kl]), 5e-6, "[-mean ll, mean KL]")
    close(resbd, resb, 5e-6, "res-bar"); close(p4bd, p4b, 1e-5, "p4-bar")
    close(p2bd, p2b, 1e-5, "p2-bar"); close(p1bd, p1b, 1e-5, "p1-bar")
    # per layer of the GRU: a wrong block would hide behind the largest one
    o = 0
    for name, n in (("Wu1", 175 * 40 + 40), ("Wu2", 40 * 50 + 50), ("Wr1", 175 * 40 + 40), ("Wr2", 40 * 50 + 50), ("Wn1", 175 * 40 + 40), ("Wn2", 40 * 100 + 100)):
        close(p1bd[o:o + n], p1b[o:o + n], 1e-5, "p1-bar " + name)
        o += n


def test_latent_handle_refuses_out_of_order_calls_and_bad_shapes():
    import torch
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    cfg = _lib.LatentConfig(max_batch=8, max_T=70, device=0)
    assert L.rnde_latent_create(C.byref(cfg), C.byref(h)) == _lib.BAD_ARG and b"max_T" in L.rnde_latent_last_error(None)
    cfg = _lib.LatentConfig(max_batch=8, max_T=5, device=0)
    _lib.check_latent(None, L.rnde_latent_create(C.byref(cfg), C.byref(h)))
    t = torch.zeros(64, device="cuda")
    assert L.rnde_latent_encode_backward(h, t.data_ptr(), 1.0, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None) == 6      # RNDE_ERR_NO_TAPE
    assert L.rnde_latent_encode(h, t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 9, 5, t.data_ptr(), t.data_ptr(), t.data_ptr(), None) == _lib.BAD_ARG
    L.rnde_latent_destroy(h)


@pytest.mark.parametrize("func,agg,lam_r", [("error_est", "mean", 50.0), ("stiff_est", "max", 10.0), ("error_stiff_est", "mean", 10.0)])
def test_fused_latent_training_step_matches_the_autograd_form(func, agg, lam_r):
    """(Parametrised over the three `save_func`s / aggregators of experiments/latent_ode.jl:153-190: EEst*dt with mean, the stiffness estimate with
    `maximum`, their blend with mean.)
    `fused_latent_loss_and_grad` (every piece of loss_function, latent_ode.jl:206-236, and of its reverse through the C ABI) against
    `latent_loss_function` + torch.autograd on the same model, data and reparameterisation sample -- the layer call in the middle is the
    same device solve on both sides, so this checks the plumbing between the five library calls: loss terms 1e-5, gradients of the four
    parameter groups 2e-4 of their largest entry (tol 1e-3 on the solve: its step sequence is not rounding noise)."""
    import torch
    import regneuralde_jl_amd as rn
    g = torch.Generator().manual_seed(5)
    B, T = 48, 49
    grid = torch.linspace(0, 1, T)
    model = rn.build_latent_ode(saveat=grid, regularize=True, generator=g, solver="Tsit5" if func == "error_est" else "AutoTsit5", reltol=1e-3, abstol=1e-3, max_batch=B, max_attempts=128)
    data = torch.randn(B, T, 37, generator=g).cuda()
    mask = (torch.rand(B, T, 37, generator=g) < 0.3).float().cuda()
    mask[:, 0, 0] = 1.0
    t_row = torch.full((B, T, 1), 1.0 / (T - 1)).cuda(); t_row[:, -1] = 0.0
    eps = torch.randn(B, 20, generator=g).cuda()

    class _Gen:                      # hands the SAME sample to the autograd form (it draws with torch.randn(..., generator=))
        pass
    real_randn = torch.randn
    try:
        torch.randn = lambda *a, **k: eps.clone()
        total_a, nll_a, kl_a, reg_a, nfe_a = rn.latent_loss_function(data, mask, t_row, model, lam_r=lam_r, lam_k=0.3, func=func, agg=torch.max if agg == "max" else torch.mean)
    finally:
        torch.randn = real_randn
    total_a.backward()
    ga = [p.grad.detach().clone() for p in model.trainable()]
    for p in model.trainable():
        p.grad = None
    total_f, nll_f, kl_f, reg_f, nfe_f = rn.fused_latent_loss_and_grad(model, data, mask, t_row, lam_r=lam_r, lam_k=0.3, eps=eps, func=func, agg=agg)
    torch.cuda.synchronize()
    assert nfe_f == nfe_a
    assert abs(float(nll_f) - float(nll_a)) <= 1e-5 * abs(float(nll_a)) and abs(float(kl_f) - float(kl_a)) <= 1e-5 * abs(float(kl_a)) + 1e-7
    # (EEst * dt is an O(dt^5) cancellation: 1e-7 differences of z0 between the two GRU implementations move it by several per cent)
    assert abs(float(reg_f) - float(reg_a)) <= 0.15 * abs(float(reg_a)) + 1e-7
    for name, a, p in zip(("p1", "p2", "p3", "p4"), ga, model.trainable()):
        err = float((p.grad - a).abs().max() / a.abs().max())
        print(name, "rel err", err)
        assert torch.isfinite(p.grad).all() and err <= 2e-4, (name, err)


def test_adamax_step_matches_flux_form():
    """rnde_adamax_step (Optimiser(InvDecay(1e-5), AdaMax(0.01)), latent_ode.jl:108) against the formulas of Flux 0.11's apply! written out in
    numpy fp64, five steps on one group: 1e-6 relative."""
    import torch
    import regneuralde_jl_amd as rn
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(1000); gs = [rng.standard_normal(1000) * (0.1 ** k) for k in range(5)]
    p = torch.from_numpy(p0.astype(np.float32)).cuda()
    opt = rn.FluxAdaMax([p], gamma=1e-5, eta=0.01)
    ref, m, u, bp = p0.copy(), np.zeros(1000), np.zeros(1000), 0.9
    for k, g in enumerate(gs):
        opt.step(grads=[torch.from_numpy(g.astype(np.float32)).cuda()])
        gg = g.astype(np.float32).astype(np.float64) / (1 + 1e-5 * (k + 1))
        m = 0.9 * m + 0.1 * gg; u = np.maximum(0.999 * u, np.abs(gg))
        ref -= 0.01 / (1 - bp) * m / (u + 1e-8); bp *= 0.9
    torch.cuda.synchronize()
    assert np.abs(p.cpu().numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
