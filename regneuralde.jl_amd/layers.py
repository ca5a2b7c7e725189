"""Host-side mirrors of the Flux pieces the hot path is built from.

Dense / TDChain / MLPDynamics only describe shapes and hold initial parameters; the arithmetic
runs in librnde.so.  Layout rule (reference src/models/neural_ode.jl:12, Flux.destructure): the
flat parameter vector is [vec(W_1) column-major (out x in); b_1; vec(W_2); b_2; ...].
A Julia D x B column-major matrix is a torch tensor of shape (B, D), contiguous.
"""
import math

import torch


class Dense:
    """Flux.Dense(in, out, act): W is out x in, Glorot-uniform, zero bias (Flux 0.11 defaults)."""

    def __init__(self, n_in, n_out, act="identity", generator=None):
        self.n_in, self.n_out, self.act = n_in, n_out, act
        lim = math.sqrt(6.0 / (n_in + n_out))
        # stored as (in, out) row-major == (out x in) column-major
        self.W = (torch.rand(n_in, n_out, generator=generator) * 2 - 1) * lim
        self.b = torch.zeros(n_out)


class TDChain:
    """reference src/models/basic.jl:1-35: every layer sees vcat(x, t)."""
    time_dep = True
    pre_act = False

    def __init__(self, *layers):
        self.layers = list(layers)

    def dims(self):
        d = [self.layers[0].n_in - 1]
        for l in self.layers:
            d.append(l.n_out)
        return d


def MLPDynamics(n_in, hidden, generator=None):
    """reference experiments/mnist_node.jl:41-54: Dense(in+1, hidden, tanh) -> Dense(hidden+1, in, tanh)."""
    return TDChain(Dense(n_in + 1, hidden, "tanh", generator), Dense(hidden + 1, n_in, "tanh", generator))


class Chain:
    """Time-independent Flux.Chain of Dense layers, optional leading tanh (experiments/latent_ode.jl:113-124)."""
    time_dep = False

    def __init__(self, *layers, pre_act=False):
        self.layers = list(layers)
        self.pre_act = pre_act

    def dims(self):
        d = [self.layers[0].n_in]
        for l in self.layers:
            d.append(l.n_out)
        return d


def LatentGenDynamics(latent=20, hidden=50, depth=8, generator=None):
    """reference experiments/latent_ode.jl:113-124 (gen_dynamics): x -> tanh.(x), then `depth` Dense layers alternating
    latent -> hidden -> latent, all tanh, time independent."""
    dims = [latent if i % 2 == 0 else hidden for i in range(depth + 1)]
    return Chain(*[Dense(dims[i], dims[i + 1], "tanh", generator) for i in range(depth)], pre_act=True)


def destructure(model):
    """Flux.destructure(model)[1]: flat fp32 parameter vector."""
    parts = []
    for l in model.layers:
        parts.append(l.W.reshape(-1))
        parts.append(l.b.reshape(-1))
    return torch.cat(parts).to(torch.float32).contiguous()
