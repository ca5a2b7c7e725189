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
    with pytest.raises(NotImplementedError):
        rnde.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=True)
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
