"""The caller of the hot path in the latent-ODE experiment, mirrored so that configuration 4 runs end
to end around the device solve:

LatentGRU, single_run            reference experiments/latent_ode.jl:39-99
LatentTimeSeriesModel            reference src/models/time_series.jl:1-70
log_likelihood, kl_divergence,
loss_function, sample_tbounds    reference experiments/latent_ode.jl:188-269
Optimiser(InvDecay, AdaMax)      reference experiments/latent_ode.jl:108

The node (gen_dynamics integrated by Tsit5 with `saveat`, latent_ode.jl:137-147) is the hot path: it runs in librnde.so (chain
engine) forward and reverse.  Round 4: the recognition GRU, the two small Dense stacks, the likelihood and the KL term run there too
(`fused_latent_loss_and_grad`, include/rnde.h: rnde_latent_*: one launch for the 49 recurrent steps, one for their reverse);
the torch formulas below stay as the autograd form of the same model (and as the cross-check of the device path in the tests).

Layouts: a Julia `F x T x B` array is a torch tensor of shape (B, T, F); flat parameter vectors are what Flux.destructure
returns for the corresponding struct (fields in declaration order, each Dense as [vec(W) column-major (out x in); b]).
"""
import math

import torch

from .layers import Chain, Dense, destructure
from .node import TrackedNeuralODE

_ACTS = {"identity": lambda v: v, "tanh": torch.tanh, "sigmoid": torch.sigmoid}


def _apply_chain(chain, p, x):
    """re(p)(x) for a Chain of Dense layers; x is (B, n_in)."""
    o = 0
    if getattr(chain, "pre_act", False):
        x = torch.tanh(x)
    for l in chain.layers:
        W = p[o:o + l.n_in * l.n_out].view(l.n_in, l.n_out)        # (in, out) row-major == out x in column-major
        o += l.n_in * l.n_out
        b = p[o:o + l.n_out]
        o += l.n_out
        x = _ACTS[l.act](x @ W + b)
    return x


def _chain_len(chain):
    return sum(l.n_in * l.n_out + l.n_out for l in chain.layers)


class LatentGRU:
    """latent_ode.jl:39-63: three two-layer stacks over vcat(y_mean, y_std, x); in_dim counts the DATA rows (37), the input
    carries data, mask and one time-difference row, i.e. 2 in_dim + 1 rows."""

    def __init__(self, in_dim, h_dim, latent_dim, generator=None):
        n_in = latent_dim * 2 + in_dim * 2 + 1
        self.update_gate = Chain(Dense(n_in, h_dim, "tanh", generator), Dense(h_dim, latent_dim, "sigmoid", generator))
        self.reset_gate = Chain(Dense(n_in, h_dim, "tanh", generator), Dense(h_dim, latent_dim, "sigmoid", generator))
        self.new_state = Chain(Dense(n_in, h_dim, "tanh", generator), Dense(h_dim, latent_dim * 2, "identity", generator))
        self.in_dim, self.latent_dim = in_dim, latent_dim

    def chains(self):
        return (self.update_gate, self.reset_gate, self.new_state)          # @functor field order

    def single_run(self, p, y_mean, y_std, x):
        """latent_ode.jl:67-97.  y_*: (B, latent); x: (B, 2 in_dim + 1)."""
        n0, n1 = _chain_len(self.update_gate), _chain_len(self.reset_gate)
        pu, pr, pn = p[:n0], p[n0:n0 + n1], p[n0 + n1:]
        y_concat = torch.cat([y_mean, y_std, x], dim=1)
        update_gate = _apply_chain(self.update_gate, pu, y_concat)
        reset_gate = _apply_chain(self.reset_gate, pr, y_concat)
        concat = torch.cat([y_mean * reset_gate, y_std * reset_gate, x], dim=1)
        new_state = _apply_chain(self.new_state, pn, concat)
        new_state_mean, new_state_std = new_state[:, :self.latent_dim], new_state[:, self.latent_dim:]
        new_y_mean = (1 - update_gate) * new_state_mean + update_gate * y_mean
        new_y_std = (1 - update_gate) * new_state_std + update_gate * y_std
        # rows (size(x,1) / 2 + 1):end of the Julia input: integer division of 2 in_dim + 1 -> the mask rows and the time row
        half = x.shape[1] // 2
        mask = (x[:, half:].sum(dim=1, keepdim=True) > 0).to(x.dtype)
        return mask * new_y_mean + (1 - mask) * y_mean, mask * new_y_std + (1 - mask) * y_std

    def __call__(self, p, x):
        """latent_ode.jl:99-106: runs over the time axis BACKWARDS from zeros; returns vcat(y_mean, y_std) as (B, 2 latent)."""
        B, T = x.shape[0], x.shape[1]
        y_mean = y_std = torch.zeros(B, self.latent_dim, dtype=x.dtype, device=x.device)
        for t in range(T - 1, -1, -1):
            y_mean, y_std = self.single_run(p, y_mean, y_std, x[:, t, :])
        return torch.cat([y_mean, y_std], dim=1)


def destructure_gru(gru):
    return torch.cat([destructure(c) for c in gru.chains()])


class LatentTimeSeriesModel:
    """time_series.jl:1-70: rnn -> enc -> (mu0, log sigma^2) -> z0 = eps * exp(log sigma^2 / 2) + mu0 -> node (saveat) -> dec.
    Parameters are the four flat vectors (p1, p2, p3, p4) of Flux.trainable (time_series.jl:38)."""

    def __init__(self, rnn: LatentGRU, enc: Chain, node: TrackedNeuralODE, dec: Dense, device="cuda"):
        self.rnn, self.enc, self.node, self.dec = rnn, enc, node, Chain(dec)
        self.p1 = destructure_gru(rnn).to(device).requires_grad_(True)
        self.p2 = destructure(enc).to(device).requires_grad_(True)
        self.p3 = node.p.to(device).clone().requires_grad_(True)
        self.p4 = destructure(self.dec).to(device).requires_grad_(True)

    def trainable(self):
        return (self.p1, self.p2, self.p3, self.p4)

    def __call__(self, x, p1=None, p2=None, p3=None, p4=None, generator=None, **node_kwargs):
        """x: (B, T, 2 in_dim + 1) = vcat(data, mask, dt).  Returns (result (B, T, in_dim), mu0, log sigma^2, nfe, sv)."""
        p1 = self.p1 if p1 is None else p1
        p2 = self.p2 if p2 is None else p2
        p3 = self.p3 if p3 is None else p3
        p4 = self.p4 if p4 is None else p4
        out = _apply_chain(self.enc, p2, self.rnn(p1, x))
        latent_dim = out.shape[1] // 2
        mu0, logvar = out[:, :latent_dim], out[:, latent_dim:]
        sample = torch.randn(mu0.shape, dtype=mu0.dtype, device=mu0.device, generator=generator)   # CUDA.randn, time_series.jl:58
        z0 = sample * torch.exp(logvar / 2) + mu0
        res, nfe, sv = self.node(z0, p3, **node_kwargs)                 # (B, T, latent): the device solve
        B, T, _ = res.shape
        result = _apply_chain(self.dec, p4, res.reshape(B * T, -1)).reshape(B, T, -1)
        return result, mu0, logvar, nfe, sv


def log_likelihood(dpred, mask, sigma=0.01):
    """latent_ode.jl:192-200: Gaussian log density of the masked residual, summed over (feature, time), divided by the number of
    observed entries; one value per sample.  (As in the reference the constant terms are counted at unobserved entries too.)"""
    ll = -dpred.pow(2) / (2 * sigma ** 2) - math.log(sigma) - math.log(2 * math.pi) / 2
    return ll.sum(dim=(1, 2)) / mask.sum(dim=(1, 2))


def kl_divergence(mu, logvar):
    """latent_ode.jl:203-204 (standard normal prior): mean over the latent rows of (exp(lv) + mu^2 - 1 - lv) / 2."""
    return (torch.exp(logvar) + mu.pow(2) - 1 - logvar).mean(dim=1) / 2


def latent_loss_function(data, mask, t_row, model, p1=None, p2=None, p3=None, p4=None, lam_r=1.0e2, lam_k=1.0, regularize=True,
                         agg=torch.mean, func="error_est", saveat=None, generator=None):
    """latent_ode.jl:206-262: -mean(log likelihood - lam_k KL) + lam_r agg(sv.saveval).
    data, mask: (B, T, in_dim); t_row: (B, T, 1) (the constant `_t` row).  Returns (total, nll, kl, reg, nfe)."""
    x_ = torch.cat([data, mask, t_row], dim=2)
    result, mu0, logvar, nfe, sv = model(x_, p1, p2, p3, p4, generator=generator, func=func, saveat=saveat)
    dpred = result * mask - data * mask
    ll = log_likelihood(dpred, mask)
    kl = lam_k * kl_divergence(mu0, logvar)
    reg = lam_r * agg(sv.saveval) if regularize and sv is not None else torch.zeros((), device=data.device)
    total = -(ll - kl).mean() + reg
    return total, (-ll.mean()).detach(), kl.mean().detach(), reg.detach(), nfe


def lambda_k(epoch):
    """latent_ode.jl:178: KL weight warm-up max(0, 1 - 0.99^(epoch - 10))."""
    return max(0.0, 1.0 - 0.99 ** (epoch - 10))


def sample_tbounds(t, dt=None, generator=None):
    """STEER on the save grid, latent_ode.jl:180-189: every time but the first moves uniformly within half its gap to the
    previous one, clamped to [0, 1].  t: 1-D tensor.  Returns (sampled, gaps)."""
    if dt is None:
        dt = t[1:] - t[:-1] + torch.finfo(torch.float32).eps
    r = torch.rand(dt.shape, dtype=dt.dtype, device=dt.device, generator=generator)
    return torch.cat([t[:1], t[1:] + (2 * r - 1) * dt / 2]).clamp_(0.0, 1.0), dt


def get_t_saveat(t, saveat, steer=False, gaps=None, generator=None):
    """latent_ode.jl:323-333.  t: (B, T, 1) observation times of the batch; saveat: 1-D grid of the layer.  Returns
    (t, save grid for this call, the time-difference row `_t` of shape (B, T, 1) with a trailing zero)."""
    if steer:
        tt, _ = sample_tbounds(saveat, gaps, generator)
        t = tt.reshape(1, -1, 1).repeat(t.shape[0], 1, 1)
    else:
        t, tt = t.to(torch.float32), saveat
    dt_row = torch.cat([t[:, 1:] - t[:, :-1], torch.zeros_like(t[:, :1])], dim=1)
    return t, tt, dt_row


class FluxAdaMax:
    """Flux.Optimise.Optimiser(InvDecay(gamma), AdaMax(eta, (0.9, 0.999))) group by group (latent_ode.jl:108; utils.jl:149-156)."""

    def __init__(self, params, gamma=1.0e-5, eta=0.01, beta=(0.9, 0.999), eps=1.0e-8):
        self.params = [p for p in params if p.numel() > 0]
        self.gamma, self.eta, self.beta, self.eps = gamma, eta, beta, eps
        self.n = [1 for _ in self.params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.u = [torch.zeros_like(p) for p in self.params]
        self.bp = [beta[0] for _ in self.params]

    @torch.no_grad()
    def step(self, grads=None):
        b1, b2 = self.beta
        for i, p in enumerate(self.params):
            g = p.grad if grads is None else grads[i]
            if g is None:
                continue
            if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous():      # one launch per group (rnde_adamax_step)
                import ctypes as C
                from . import _lib
                st = _lib.lib().rnde_adamax_step(p.data_ptr(), g.data_ptr(), self.m[i].data_ptr(), self.u[i].data_ptr(), p.numel(), self.n[i], self.gamma,
                                                 self.eta, b1, b2, self.eps, self.bp[i], C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
                if st != 0:
                    raise _lib.RndeError(st, "rnde_adamax_step")
                self.n[i] += 1
                self.bp[i] *= b1
                p.grad = None
                continue
            g = g / (1.0 + self.gamma * self.n[i])                       # InvDecay
            self.n[i] += 1
            self.m[i].mul_(b1).add_(g, alpha=1 - b1)
            torch.maximum(self.u[i] * b2, g.abs(), out=self.u[i])
            p.sub_((self.eta / (1 - self.bp[i])) * self.m[i] / (self.u[i] + self.eps))
            self.bp[i] *= b1
            p.grad = None


def build_latent_ode(in_dim=37, h_dim=40, rec_dim=50, latent=20, hidden=50, depth=8, saveat=None, regularize=True,
                     generator=None, device="cuda", solver="Tsit5", **solver_kwargs):
    """The model of latent_ode.jl:111-147 at the reference's sizes: LatentGRU(37, 40, 50), rec_to_gen
    Dense(100, 50, tanh) -> Dense(50, 40), gen_dynamics (tanh + 8 Dense 20 <-> 50), gen_to_data Dense(20, 37)."""
    from .layers import LatentGenDynamics
    rnn = LatentGRU(in_dim, h_dim, rec_dim, generator)
    enc = Chain(Dense(2 * rec_dim, rec_dim, "tanh", generator), Dense(rec_dim, 2 * latent, "identity", generator))
    dyn = LatentGenDynamics(latent, hidden, depth, generator)
    kw = dict(reltol=1.4e-8, abstol=1.4e-8)
    kw.update(solver_kwargs)
    node = TrackedNeuralODE(dyn, [0.0, 1.0], False, regularize, solver, saveat=saveat, **kw)      # solver: "AutoTsit5" for the stiffness callbacks (latent_ode.jl:127-136)
    dec = Dense(latent, in_dim, "identity", generator)
    return LatentTimeSeriesModel(rnn, enc, node, dec, device=device)


class _LatentHandle:
    def __init__(self, max_batch, max_T, device_index):
        import ctypes as C
        from . import _lib
        self.L = _lib.lib()
        self.ptr = C.c_void_p()
        cfg = _lib.LatentConfig(max_batch=max_batch, max_T=max_T, device=device_index)
        _lib.check_latent(None, self.L.rnde_latent_create(C.byref(cfg), C.byref(self.ptr)))
        self.max_batch, self.max_T = max_batch, max_T

    def __del__(self):
        try:
            if self.ptr:
                self.L.rnde_latent_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


def fused_latent_loss_and_grad(model, data, mask, t_row, lam_r=1.0e2, lam_k=1.0, regularize=True, saveat=None, generator=None, eps=None, func=None, agg="mean"):
    """One training-step gradient of the latent-ODE model WITHOUT a tape library in the loop: every piece of
    loss_function (experiments/latent_ode.jl:206-236) and of its reverse runs in librnde.so --

        rnde_latent_encode          recognition GRU (49 steps, one launch), rec_to_gen, z0 = eps * exp(logvar / 2) + mu0, KL
        rnde_node_forward_saveat    the layer call (the hot path; taped)
        rnde_latent_decode_loss     gen_to_data, masked likelihood, their reverse
        rnde_node_backward_async    reverse sweep of the solve
        rnde_latent_encode_backward reverse of rec_to_gen and of the GRU (one launch) + the weight-gradient GEMMs

    Same loss surface as `latent_loss_function`; `func` (closure or name: latent_ode.jl:154-190 defines the same three `save_func`s as mnist_node.jl) and
    `agg` ("mean" / "max": `maximum` for stiff_est, :171) as for `classifier.fused_loss_and_grad`.  Sets .grad on (p1, p2, p3, p4); returns
    (total, nll, kl, reg, nfe) -- total / nll / kl as device tensors (nothing but the solver's step log is read on the host).
    data, mask: (B, T, in_dim); t_row: (B, T, 1).  eps: the standard-normal sample (B, latent) (default: drawn here, CUDA.randn of time_series.jl:58)."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    dev = data.device
    B, T, _ = data.shape
    x_ = torch.cat([data, mask, t_row], dim=2).to(torch.float32).contiguous()
    node = model.node
    lat = 20                                     # latent state rows (latent_ode.jl:112-124); the kernels are built for the reference's sizes
    # rnde_latent_* is compiled for the experiment's own sizes (latent_ode.jl:39-124: LatentGRU(37, 40, 50), rec_to_gen 100-50-40, gen_to_data 20-37):
    # a model built with other sizes must not reach kernels that would read its parameter vectors out of bounds
    # (checked on EVERY call: the widths belong to the call's arrays, the parameter vectors can be replaced between calls; only the library's three counts are cached)
    want_n = getattr(model, "_latent_want", None)
    if want_n is None:
        n1, n2, n4 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.rnde_latent_param_counts(C.byref(n1), C.byref(n2), C.byref(n4))
        want_n = model._latent_want = (n1.value, n2.value, n4.value)
    p1_, p2_, p3_, p4_ = model.trainable()
    have = (p1_.numel(), p2_.numel(), p4_.numel(), data.shape[2], mask.shape[2], t_row.shape[2], node.model.dims()[0])
    want = want_n + (37, 37, 1, lat)
    if have != want:
        raise ValueError("fused_latent_loss_and_grad runs the kernels built for the reference's latent-ODE sizes (experiments/latent_ode.jl:39-124): "
                         f"(len p1, len p2, len p4, data width, mask width, time width, latent rows) must be {want}, this model has {have}; "
                         "use latent_loss_function (torch autograd around the layer call) for other sizes")
    hl = getattr(model, "_latent_handle", None)
    if hl is None or hl.max_batch < B or hl.max_T < T:
        hl = model._latent_handle = _LatentHandle(max(B, getattr(hl, "max_batch", 0) if hl else 0), max(T, getattr(hl, "max_T", 0) if hl else 0), dev.index or 0)
    if eps is None:
        eps = torch.randn(B, lat, dtype=torch.float32, device=dev, generator=generator)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p1, p2, p3, p4 = (p.detach() for p in model.trainable())
    z0 = torch.empty(B, lat, dtype=torch.float32, device=dev)
    mu0, logvar = torch.empty_like(z0), torch.empty_like(z0)
    _tr = getattr(model, "_trace", None)      # (diagnostics: host time of each library call of this step, appended when the caller set model._trace = [])
    if _tr is not None:
        import time as _time
        _t = [_time.perf_counter()]
        _mark = lambda: _t.append(_time.perf_counter())
    else:
        _mark = lambda: None
    _lib.check_latent(hl.ptr, L.rnde_latent_encode(hl.ptr, x_.data_ptr(), p1.data_ptr(), p2.data_ptr(), eps.contiguous().data_ptr(), B, T,
                                                   z0.data_ptr(), mu0.data_ptr(), logvar.data_ptr(), stream))
    _mark()
    # the layer call on z0 (time_series.jl:61): forward with saveat, taped
    grid = node._saveat_times(node.kwargs["saveat"] if saveat is None else saveat, node.tspan)      # update_saveat!, neural_ode.jl:35-46
    if len(grid) != T:
        raise ValueError("one save time per observation time (latent_ode.jl:137)")
    node._func = node.resolve_func(func)
    agg_max = agg in ("max", "maximum", torch.max)
    if not agg_max and agg not in ("mean", torch.mean):
        raise ValueError("agg: 'mean' or 'max' (latent_ode.jl:153,:171)")
    hn = node._acquire(z0, True)
    res = torch.empty(B, T, lat, dtype=torch.float32, device=dev)
    sa = (C.c_float * T)(*grid)
    sv_host = (C.c_float * (node.max_attempts + 1))()
    nfe, nsv = C.c_int64(0), C.c_int32(0)
    _lib.check(hn.ptr, L.rnde_node_forward_saveat(hn.ptr, z0.data_ptr(), p3.data_ptr(), B, node.tspan[0], node.tspan[1], sa, T, res.data_ptr(),
                                                  C.byref(nfe), sv_host, C.byref(nsv), 1, stream))
    _mark()
    loss2 = torch.empty(2, dtype=torch.float32, device=dev)
    resb = torch.empty_like(res)
    p4bar = torch.empty_like(p4)
    _lib.check_latent(hl.ptr, L.rnde_latent_decode_loss(hl.ptr, res.data_ptr(), p4.data_ptr(), x_.data_ptr(), B, T, loss2.data_ptr(), resb.data_ptr(),
                                                       p4bar.data_ptr(), stream))
    _mark()
    n = nsv.value
    reg = 0.0
    svb = None
    if regularize and node.regularize and n > 0:
        if agg_max:                                                          # lam_r * maximum(sv.saveval)
            vals = [sv_host[i] for i in range(n)]
            k = max(range(n), key=lambda i: vals[i])
            reg = lam_r * vals[k]
            svb = (C.c_float * n)(*[lam_r if i == k else 0.0 for i in range(n)])
        else:
            reg = lam_r * sum(sv_host[i] for i in range(n)) / n             # lam_r * mean(sv.saveval), latent_ode.jl:233
            svb = (C.c_float * n)(*([lam_r / n] * n))
    z0bar, p3bar = torch.empty_like(z0), torch.empty_like(p3)
    _lib.check(hn.ptr, L.rnde_node_backward_async(hn.ptr, resb.data_ptr(), svb, z0bar.data_ptr(), p3bar.data_ptr(), None, stream))
    _mark()
    p1bar, p2bar = torch.empty_like(p1), torch.empty_like(p2)
    _lib.check_latent(hl.ptr, L.rnde_latent_encode_backward(hl.ptr, z0bar.data_ptr(), float(lam_k), p1.data_ptr(), p2.data_ptr(), x_.data_ptr(),
                                                           p1bar.data_ptr(), p2bar.data_ptr(), stream))
    _mark()
    if _tr is not None:
        _tr.append([round(1e3 * (b - a), 3) for a, b in zip(_t, _t[1:])])      # [encode, layer forward (host wait inside), decode + loss, layer reverse (async), encode reverse]
    model.p1.grad, model.p2.grad, model.p3.grad, model.p4.grad = p1bar, p2bar, p3bar, p4bar
    node.last_nfe = int(nfe.value)
    nll, kl = loss2[0], lam_k * loss2[1]
    return nll + kl + reg, nll, kl, reg, int(nfe.value)
