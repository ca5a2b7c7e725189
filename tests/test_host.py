"""Host-side mirror of the reference interface (no GPU): layout, loss, optimiser, schedule, error behaviour."""
import math

import numpy as np
import pytest
import torch


def test_destructure_layout_matches_oracle_convention(rnde):
    from oracle.oracle import Oracle, arch_mnist
    g = torch.Generator().manual_seed(0)
    dyn = rnde.MLPDynamics(12, 5, generator=g)
    dyn.layers[0].b.uniform_(-0.1, 0.1, generator=g)
    dyn.layers[1].b.uniform_(-0.1, 0.1, generator=g)
    p = rnde.destructure(dyn).numpy()
    assert p.shape == ((12 + 1) * 5 + 5 + (5 + 1) * 12 + 12,)
    u = torch.rand(3, 12, generator=g); t = 0.3
    W1, b1, W2, b2 = dyn.layers[0].W, dyn.layers[0].b, dyn.layers[1].W, dyn.layers[1].b   # stored (in, out)
    tt = torch.full((3, 1), t)
    h = torch.tanh(torch.cat([u, tt], 1) @ W1 + b1)
    ref = torch.tanh(torch.cat([h, tt], 1) @ W2 + b2)                                        # mnist_node.jl:51-54
    got = Oracle(arch_mnist(12, 5), np.float64).f_eval(p.astype(np.float64), u.numpy().astype(np.float64), t)
    np.testing.assert_allclose(got, ref.numpy(), atol=1e-6)


def test_logitcrossentropy_matches_torch(rnde):
    g = torch.Generator().manual_seed(1)
    pred = torch.randn(7, 10, generator=g)
    lab = torch.randint(0, 10, (7,), generator=g)
    y = torch.eye(10)[lab]
    assert torch.allclose(rnde.logitcrossentropy(pred, y), torch.nn.functional.cross_entropy(pred, lab), atol=1e-6)


def test_flux_optimiser_invdecay_momentum(rnde):
    p = torch.tensor([1.0, -2.0], requires_grad=True)
    opt = rnde.FluxOptimiser([torch.zeros(0), p], gamma=1e-5, eta=0.1, rho=0.9)   # empty group skipped (utils.jl:151)
    ref_p, v = np.array([1.0, -2.0]), np.zeros(2)
    for n in range(1, 4):
        gnp = np.array([0.5 * n, -1.0])
        p.grad = torch.tensor(gnp, dtype=torch.float32)
        opt.step()
        d = gnp / (1 + 1e-5 * n)              # InvDecay: state starts at 1
        v = 0.9 * v - 0.1 * d                 # Momentum
        ref_p = ref_p + v
        np.testing.assert_allclose(p.detach().numpy(), ref_p, rtol=1e-6)


def test_lambda_schedule_endpoints(rnde):
    assert rnde.lambda_schedule(0) == pytest.approx(100.0)
    assert rnde.lambda_schedule(75) == pytest.approx(10.0)
    assert rnde.lambda_schedule(37.5) == pytest.approx(math.sqrt(1000.0))


def test_node_constructor_contract(rnde):
    dyn = rnde.MLPDynamics(784, 100)
    node = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1.4e-8, abstol=1.4e-8,
                                 save_start=False)
    assert node.P == 158568 and node.return_multiple is False and node.regularize
    multi = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", saveat=[0.0, 0.5, 1.0])
    assert multi.return_multiple is True                      # neural_ode.jl:11
    every = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=True)
    assert every.return_multiple is True and every.save_everystep is True      # the other way to return_multiple (neural_ode.jl:10-11)
    assert rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=True, saveat=[0.0, 1.0]).save_everystep is False   # saveat decides
    assert rnde.TrackedNeuralODE._saveat_times(0.25, [0.0, 1.0]) == [0.0, 0.25, 0.5, 0.75, 1.0]
    assert rnde.TrackedNeuralODE._saveat_times(0.4, [0.0, 1.0]) == pytest.approx([0.0, 0.4, 0.8, 1.0])
    with pytest.raises(ValueError):
        rnde.TrackedNeuralODE._saveat_times([0.5, 0.2], [0.0, 1.0])
    with pytest.raises(ValueError):
        rnde.TrackedNeuralODE._saveat_times([0.5, 1.2], [0.0, 1.0])
    with pytest.raises(ValueError):
        rnde.TrackedNeuralODE(dyn, [0.0, 1.0], False, True, "Tsit5")
    with pytest.raises(ValueError):
        rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Vern7")


def test_latent_gen_dynamics_shape(rnde):
    dyn = rnde.LatentGenDynamics()
    assert dyn.dims() == [20, 50, 20, 50, 20, 50, 20, 50, 20] and dyn.pre_act and not dyn.time_dep
    assert rnde.destructure(dyn).numel() == 8280                   # latent ODE P = 8,280


def test_node_refuses_cpu_tensors_loudly(rnde):
    dyn = rnde.MLPDynamics(8, 4)
    node = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, False, "Tsit5", reltol=1e-3, abstol=1e-3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        node(torch.rand(2, 8))


def test_shard_columns(rnde):
    x = torch.arange(20).reshape(10, 2)
    a, b = rnde.shard_columns(x, 0, 2), rnde.shard_columns(x, 1, 2)
    assert torch.equal(torch.cat([a, b]), x) and a.shape[0] == 5


def test_regulariser_table_and_steer(rnde):
    lam0, lam1, func, agg, solver = rnde.REGULARISERS["stiff_est"]
    assert (lam0, lam1, func, solver) == (0.1, 0.1, "stiff_est", "AutoTsit5") and agg is torch.max      # mnist_node.jl:70-83
    assert rnde.REGULARISERS["error_est"][:2] == (1.0e2, 1.0e1) and rnde.REGULARISERS["error_stiff_est"][:2] == (1.0e1, 1.0e1)
    g = torch.Generator().manual_seed(0)
    ts = [rnde.sample_tspan_ubound(generator=g) for _ in range(200)]
    assert all(t0 == 0.0 and 0.5 <= t1 <= 1.5 for t0, t1 in ts)                                          # mnist_node.jl:104-105
    assert max(t1 for _, t1 in ts) > 1.3 and min(t1 for _, t1 in ts) < 0.7
    dyn = rnde.MLPDynamics(8, 4)
    node = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "AutoTsit5", reltol=1e-3, abstol=1e-3)     # test_node.jl:60-72
    assert node.regularize


def test_callers_func_is_recognised_not_replaced(rnde):
    """The reference hands the layer a closure (`model(x, p1, p2, p3; func = save_func, ...)`, experiments/mnist_node.jl:134).  The three
    `save_func`s of mnist_node.jl:62-103 / mnist_nsde.jl:45-61, restated here as Python closures, must select the regulariser they compute --
    and a closure the kernels do not compute must raise, never train on another loss."""
    from regneuralde_jl_amd import node as N
    stab = 1.0 / 3.5068                                             # `stability_size` of mnist_node.jl:72-73
    err = lambda u, t, integ: integ.EEst * integ.dt                  # :67
    def stiff(u, t, integ):                                          # :74-79
        s = abs(integ.eigen_est)
        return stab * (0 if (s == 0 or math.isnan(s)) else s)
    def both(u, t, integ):                                           # :88-97
        e, s = integ.EEst * integ.dt, integ.eigen_est
        return ((0 if e == 0 else e) + 0.1 * stab * (0 if s == 0 else s)) * 1.0
    assert N.reg_code(lambda u, t, integ: 0, N.TSIT5_STABILITY_SIZE) == 0          # neural_ode.jl:54
    assert N.reg_code(err, N.TSIT5_STABILITY_SIZE) == 1
    assert N.reg_code(stiff, N.TSIT5_STABILITY_SIZE) == 2
    assert N.reg_code(both, N.TSIT5_STABILITY_SIZE) == 3
    assert N.reg_code(lambda u, t, integ: torch.tensor(integ.EEst) * integ.dt, 3.5068) == 1       # tracked scalars are fine
    assert N.reg_code(lambda u, t, integ: abs(integ.eigen_est * integ.dt), N.TSIT5_STABILITY_SIZE) == 4      # the reference's own test: test/test_node.jl:75,:84
    sde_stiff = lambda u, t, integ: abs(integ.eigen_est) / 10.6      # mnist_nsde.jl:53-58
    assert N.reg_code(sde_stiff, N.SOSRI2_STABILITY_SIZE) == 2
    for bad in (lambda u, t, integ: 2 * integ.EEst * integ.dt, lambda u, t, integ: integ.EEst, sde_stiff):
        with pytest.raises(ValueError, match="none of the callbacks"):
            N.reg_code(bad, N.TSIT5_STABILITY_SIZE)
    # what the reference's run records under a plain solver: eigen_est stays 0 there
    assert N.effective_reg(3, composite=True) == 3 and N.effective_reg(1, False) == 1 and N.effective_reg(0, False) == 0
    for code in (2, 3, 4):      # (3, the blend: eigen_est's initial value under a plain solver is 1, not 0 -- it is refused like the others, not mapped to EEst*dt)
        with pytest.raises(ValueError, match="composite"):
            N.effective_reg(code, composite=False)
    assert N.effective_reg(4, composite=True) == 4
    # the layer resolves the closure before it looks for a handle (no GPU needed to get that far: the cuda check comes first)
    dyn = rnde.MLPDynamics(8, 4)
    node = rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "AutoTsit5", reltol=1e-3, abstol=1e-3)
    with pytest.raises(RuntimeError, match="cuda"):
        node(torch.zeros(2, 8), func=stiff)


def test_julia_patches_dispatch_on_the_callers_func():
    """bindings/julia cannot run here; what can be checked is the text: every regularised call method derives the handle's `regularize` code
    from `func` (no side keyword the unchanged experiment script would never pass), the handle tables are keyed by the code, the SDE patch reads
    the solver off the layer, and RNDE.jl's rule is the one node.py implements (same probes, same four candidates)."""
    import os
    import re
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bindings", "julia")
    ode, sde, mod = (open(os.path.join(d, f)).read() for f in ("patch_neural_ode.jl", "patch_neural_sde.jl", "RNDE.jl"))
    assert "kind::Symbol" not in ode and "kind =" not in ode
    # a layer the library cannot represent is refused, never skipped (experiments/sde_toy_problem.jl's drift starts with `x -> x .^ 3`)
    assert "all(l -> l isa Flux.Dense, ds) || error(" in ode and "all(l -> l isa Flux.Dense, ls) || error(" in sde and "_is_tanh_layer(first(layers))" in ode
    assert len(re.findall(r"h = rnde_handle\(n, size\(x, 2\), _reg_code\(n, func\)\)", ode)) == 4
    assert len(re.findall(r"h = rnde_handle\(n, size\(x, 2\), func\)", sde)) == 4
    assert "get!(tab, (B, code))" in ode and "get!(tab, (B, reg))" in sde
    assert "RNDE.solver_name(n.args)" in ode and "RNDE.solver_name(n.args)" in sde and "solver = solver" in sde
    assert "MockIntegrator(2f0, 3f0, 5f0), MockIntegrator(0.5f0, 0.25f0, -7f0)" in mod
    from regneuralde_jl_amd import node as N
    assert N._PROBES == ((2.0, 3.0, 5.0), (0.5, 0.25, -7.0))
    assert "TSIT5_STABILITY_SIZE = 3.5068" in ode and "SOSRI2_STABILITY_SIZE = 10.6" in sde
    assert (N.TSIT5_STABILITY_SIZE, N.SOSRI2_STABILITY_SIZE) == (3.5068, 10.6)
    for code in ("REG_NONE", "REG_ERR", "REG_STIFF", "REG_ERR_STIFF", "REG_STIFF_DT"):
        assert re.search(code + r" => m ->", mod), code


# ---- latent time-series caller (experiments/latent_ode.jl, src/models/time_series.jl) ---------------------------------------

def _dense_cm(p, o, n_in, n_out):
    """One Dense out of a Flux.destructure vector, read the JULIA way: W = reshape(p[o+1 : o+out*in], out, in) column-major."""
    W = p[o:o + n_in * n_out].reshape(n_out, n_in, order="F")
    b = p[o + n_in * n_out:o + n_in * n_out + n_out]
    return W, b, o + n_in * n_out + n_out


def test_latent_gru_matches_a_column_major_restatement(rnde):
    """LatentGRU against a loop written in the reference's own orientation (features x batch, W out x in taken column-major from
    the flat vector): pins the parameter layout of p1 and the backwards time loop / mask rule of latent_ode.jl:67-106."""
    g = torch.Generator().manual_seed(4)
    in_dim, h_dim, lat, B, T = 5, 7, 6, 4, 9
    gru = rnde.LatentGRU(in_dim, h_dim, lat, generator=g)
    for c in gru.chains():
        for l in c.layers:
            l.b.uniform_(-0.2, 0.2, generator=g)
    from regneuralde_jl_amd.timeseries import destructure_gru
    p = destructure_gru(gru)
    n_in = 2 * lat + 2 * in_dim + 1
    assert p.numel() == 2 * (n_in * h_dim + h_dim + h_dim * lat + lat) + n_in * h_dim + h_dim + h_dim * 2 * lat + 2 * lat
    data = torch.randn(B, T, in_dim, generator=g)
    mask = (torch.rand(B, T, in_dim, generator=g) > 0.6).float()
    mask[:, 3] = 0                                   # a time with nothing observed anywhere ...
    dt = torch.rand(B, T, 1, generator=g); dt[:, 3] = 0   # ... and a zero time row: the state must pass through unchanged
    x = torch.cat([data, mask, dt], 2)
    got = gru(p, x).numpy()

    pn = p.numpy().astype(np.float64); o = 0
    Ws = []
    for n_out2 in (lat, lat, 2 * lat):                           # update_gate, reset_gate, new_state
        W1, b1, o = _dense_cm(pn, o, n_in, h_dim)
        W2, b2, o = _dense_cm(pn, o, h_dim, n_out2)
        Ws.append((W1, b1, W2, b2))
    sig = lambda v: 1 / (1 + np.exp(-v))
    ym = ys = np.zeros((lat, B))
    xj = x.numpy().astype(np.float64).transpose(2, 1, 0)          # F x T x B
    for t in range(T - 1, -1, -1):
        xt = xj[:, t, :]
        yc = np.vstack([ym, ys, xt])
        ug = sig(Ws[0][2] @ np.tanh(Ws[0][0] @ yc + Ws[0][1][:, None]) + Ws[0][3][:, None])
        rg = sig(Ws[1][2] @ np.tanh(Ws[1][0] @ yc + Ws[1][1][:, None]) + Ws[1][3][:, None])
        cc = np.vstack([ym * rg, ys * rg, xt])
        ns = Ws[2][2] @ np.tanh(Ws[2][0] @ cc + Ws[2][1][:, None]) + Ws[2][3][:, None]
        nm, nsd = (1 - ug) * ns[:lat] + ug * ym, (1 - ug) * ns[lat:] + ug * ys
        m = (xt[xt.shape[0] // 2:].sum(0, keepdims=True) > 0).astype(np.float64)   # Julia rows (75 / 2 + 1):end, 0-based 37:
        ym, ys = m * nm + (1 - m) * ym, m * nsd + (1 - m) * ys
    np.testing.assert_allclose(got, np.vstack([ym, ys]).T, atol=2e-6)


def test_latent_likelihood_and_kl_formulas(rnde):
    g = torch.Generator().manual_seed(6)
    d = torch.randn(3, 5, 4, generator=g) * 0.02
    m = (torch.rand(3, 5, 4, generator=g) > 0.5).float(); m[0, 0, 0] = 1
    ll = rnde.log_likelihood(d * m, m).numpy()
    ref = [(-(d[i] * m[i]).numpy() ** 2 / (2 * 0.01 ** 2) - math.log(0.01) - math.log(2 * math.pi) / 2).sum() / m[i].sum().item()
           for i in range(3)]
    np.testing.assert_allclose(ll, ref, rtol=1e-5)
    mu, lv = torch.randn(3, 6, generator=g), torch.randn(3, 6, generator=g)
    kl = rnde.kl_divergence(mu, lv).numpy()
    np.testing.assert_allclose(kl, ((np.exp(lv.numpy()) + mu.numpy() ** 2 - 1 - lv.numpy()).mean(1)) / 2, rtol=1e-5)
    assert (rnde.kl_divergence(torch.zeros(2, 6), torch.zeros(2, 6)) == 0).all()
    assert rnde.lambda_k(10) == 0.0 and rnde.lambda_k(5) == 0.0 and abs(rnde.lambda_k(11) - 0.01) < 1e-12


def test_flux_adamax_recurrence(rnde):
    p = torch.tensor([1.0, -2.0, 0.5], requires_grad=True)
    opt = rnde.FluxAdaMax([torch.zeros(0), p], gamma=1e-5, eta=0.01)
    ref, m, u, bp = p.detach().clone().double(), torch.zeros(3).double(), torch.zeros(3).double(), 0.9
    for k in range(4):
        grad = torch.tensor([0.3, -0.1 * (k + 1), 0.0])
        p.grad = grad.clone()
        opt.step()
        gg = grad.double() / (1 + 1e-5 * (k + 1))
        m = 0.9 * m + 0.1 * gg
        u = torch.maximum(0.999 * u, gg.abs())
        ref = ref - (0.01 / (1 - bp)) * m / (u + 1e-8)
        bp *= 0.9
    np.testing.assert_allclose(p.detach().numpy(), ref.numpy(), rtol=1e-6)


def test_sample_tbounds_and_time_row(rnde):
    g = torch.Generator().manual_seed(8)
    grid = torch.linspace(0, 1, 49)
    tt, gaps = rnde.sample_tbounds(grid, generator=g)
    assert tt[0] == 0 and tt.min() >= 0 and tt.max() <= 1 and gaps.shape == (48,)
    assert ((tt[1:] - grid[1:]).abs() <= gaps / 2 + 1e-7).all()
    assert (tt[1:] > tt[:-1]).all()                       # half-gap jitter cannot reorder the grid
    t = grid.reshape(1, -1, 1).repeat(3, 1, 1)
    t2, tt2, row = rnde.get_t_saveat(t, grid)
    assert tt2 is grid and row.shape == (3, 49, 1) and row[:, -1].abs().max() == 0
    np.testing.assert_allclose(row[0, :-1, 0].numpy(), (grid[1:] - grid[:-1]).numpy(), atol=1e-7)
    t3, tt3, row3 = rnde.get_t_saveat(t, grid, steer=True, gaps=gaps, generator=g)
    assert t3.shape == (3, 49, 1) and torch.equal(t3[0, :, 0], tt3)


def test_shard_columns_equal_shards_for_the_coupled_controller():
    """contiguous column blocks; the coupled controller (mode 2) needs equal shards -- a ragged split is refused, not
    silently handed to an element-wise all-reduce of different lengths."""
    import pytest
    import torch
    from regneuralde_jl_amd.dataparallel import shard_columns
    x = torch.arange(10 * 3).reshape(10, 3)
    parts = [shard_columns(x, r, 4) for r in range(4)]
    assert [p.shape[0] for p in parts] == [3, 3, 3, 1] and torch.equal(torch.cat(parts), x)
    assert [shard_columns(x, r, 5, equal=True).shape[0] for r in range(5)] == [2] * 5
    with pytest.raises(ValueError):
        shard_columns(x, 0, 4, equal=True)


def test_latent_oracle_gradients_match_finite_differences():
    """oracle/latent_oracle.py (the fp64 numpy restatement of LatentGRU, rec_to_gen + sampling, gen_to_data + the masked likelihood and KL,
    reference experiments/latent_ode.jl:39-106, :192-204, src/models/time_series.jl:40-70) is what the HIP kernels of the latent-ODE caller are
    checked against: its hand-written reverse passes are pinned here by central differences of the scalar loss in fp64 (1e-6 relative), and its
    forward against the torch mirror of the same formulas (regneuralde.jl_amd/timeseries.py)."""
    import torch
    from oracle import latent_oracle as lo
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import timeseries as ts
    rng = np.random.default_rng(3)
    B, T = 5, 6
    S = lo.GruShape(4, 6, 5)                      # small shapes: in_dim 4 (9 input rows), h 6, latent 5
    x = rng.standard_normal((B, T, S.nx))
    x[:, :, 4:8] = (rng.uniform(size=(B, T, 4)) < 0.4)          # mask rows 0 / 1; some steps unobserved
    x[:, 2, 4:] = 0.0                                              # a step whose mask AND time rows are zero: the state passes through
    p1 = 0.4 * rng.standard_normal(S.n_params())
    p2 = 0.4 * rng.standard_normal(10 * 7 + 7 + 7 * 6 + 6)       # rec_to_gen: Dense(10, 7, tanh) -> Dense(7, 6): latent 3
    p4 = 0.4 * rng.standard_normal(3 * 4 + 4)                    # gen_to_data: Dense(3, 4)
    eps = rng.standard_normal((B, 3))
    data = rng.standard_normal((B, T, 4)); mask = x[:, :, 4:8].copy(); mask[:, 0, 0] = 1.0
    Wz = 0.3 * rng.standard_normal((3, T * 3))                   # a stand-in for the solve: res = tanh(z0 Wz), linear enough to differentiate by hand

    def total(p1, p2, p4):
        y, tape1 = lo.gru_forward(S, p1, x)
        z0, mu0, lv, tape2 = lo.encode_forward(p2, y, eps, rec=7, latent=3)
        res = np.tanh(z0 @ Wz).reshape(B, T, 3)
        nll, resb, p4b, _ = lo.decode_loss(p4, res, data, mask)
        loss = nll + 0.7 * lo.kl_per_sample(mu0, lv).mean()
        return loss, (tape1, tape2, res, resb, p4b)

    loss, (tape1, tape2, res, resb, p4b) = total(p1, p2, p4)
    z0b = ((resb.reshape(B, -1) * (1 - res.reshape(B, -1) ** 2)) @ Wz.T)
    yb, p2b = lo.encode_backward(p2, tape2, z0b, 0.7 / B, rec=7, latent=3)
    p1b = lo.gru_backward(S, p1, tape1, yb)
    for (vec, grad, which) in ((p1, p1b, 0), (p2, p2b, 1), (p4, p4b, 2)):
        for i in rng.choice(len(vec), 12, replace=False):
            h = 1e-6
            a = [p1.copy(), p2.copy(), p4.copy()]; a[which][i] += h
            b = [p1.copy(), p2.copy(), p4.copy()]; b[which][i] -= h
            fd = (total(*a)[0] - total(*b)[0]) / (2 * h)
            assert abs(fd - grad[i]) <= 1e-6 * max(1.0, abs(fd)) + 2e-8, (which, i, fd, grad[i])
    # forward against the torch mirror (same formulas, written independently for round 2)
    g = torch.Generator().manual_seed(0)
    gru = ts.LatentGRU(4, 6, 5, g)
    y_t = gru(torch.from_numpy(p1), torch.from_numpy(x))
    assert np.abs(y_t.numpy() - lo.gru_forward(S, p1, x)[0]).max() <= 1e-12
    pred = rng.standard_normal(data.shape)
    ll_t = ts.log_likelihood(torch.from_numpy(pred * mask - data * mask), torch.from_numpy(mask))
    d = pred * mask - data * mask
    ll_o = (-(d * d) / (2 * lo.SIGMA ** 2) - np.log(lo.SIGMA) - np.log(2 * np.pi) / 2).sum(axis=(1, 2)) / mask.sum(axis=(1, 2))
    assert np.abs(ll_t.numpy() - ll_o).max() <= 1e-9 * np.abs(ll_o).max()


def test_fused_latent_caller_refuses_other_sizes(rnde):
    """rnde_latent_* is compiled for the reference's latent-ODE sizes (experiments/latent_ode.jl:39-124); a model of other sizes must be refused before
    any kernel reads its parameter vectors (round-4 review: the counts were exported and never checked).  No GPU needed: the check comes first."""
    from regneuralde_jl_amd import timeseries as ts
    m = ts.build_latent_ode(in_dim=12, saveat=[0.0, 0.5, 1.0], device="cpu")
    with pytest.raises(ValueError, match="reference's latent-ODE sizes"):
        ts.fused_latent_loss_and_grad(m, torch.zeros(4, 3, 12), torch.zeros(4, 3, 12), torch.zeros(4, 3, 1))
