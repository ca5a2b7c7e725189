"""Callers of the hot path, mirrored so the reference's training step runs end to end:

ClassifierNODE        reference src/models/supervised_classification.jl:2-46
loss_function         reference experiments/mnist_node.jl:132-152
Optimiser(InvDecay(1e-5), Momentum(0.1, 0.9)) + update_parameters!
                      reference experiments/mnist_node.jl:130, src/utils.jl:149-156

The pre/post layers, the loss and the optimiser are a few tiny PyTorch ops around the solve; the
integration itself (forward and reverse) is librnde.so.
"""
import math
import os

import torch

from .layers import Dense
from .node import TrackedNeuralODE


class ClassifierNODE:
    """preode (reshape to 784 x B) -> TrackedNeuralODE -> postode Dense(784, 10).  Parameters are the flat
    vectors (p1, p2, p3) exactly as Flux.destructure produces them (supervised_classification.jl:9-13)."""

    def __init__(self, node: TrackedNeuralODE, post: Dense, device="cuda"):
        self.node = node
        self.post_shape = (post.n_in, post.n_out)
        self.p1 = torch.zeros(0, device=device)                                     # reshape has no parameters
        self.p2 = node.p.to(device).clone().requires_grad_(True)
        self.p3 = torch.cat([post.W.reshape(-1), post.b]).to(device).clone().requires_grad_(True)

    def trainable(self):
        return (self.p1, self.p2, self.p3)

    def __call__(self, x, p1=None, p2=None, p3=None, **node_kwargs):
        p2 = self.p2 if p2 is None else p2
        p3 = self.p3 if p3 is None else p3
        x = x.reshape(x.shape[0], -1)                                               # Chain(x -> reshape(x, 784, :))
        u, nfe, sv = self.node(x, p2, **node_kwargs)
        n_in, n_out = self.post_shape
        W = p3[: n_in * n_out].view(n_in, n_out)                                    # (in, out) row-major == out x in col-major
        b = p3[n_in * n_out:]
        return u @ W + b, nfe, sv


def fused_loss_and_grad(model, x, y, lam=1.0e2, regularize=True, tspan=None, sync=True, flat=None, reducer=None, func=None, agg="mean"):
    """One training-step gradient without a tape library in the loop:
    [solve, taped] -> [fused Dense(784,10) + logitcrossentropy + their reverse] -> [reverse solve], all through the C ABI.
    Same loss surface as `loss_function` (experiments/mnist_node.jl:132-137): sets .grad on p2 and p3 and
    returns (total_loss, cross_entropy, reg, nfe) as Python floats / int (the call already synchronises).
    func: the experiment's `save_func` -- a closure or a name, as for the layer call (None: the layer's default EEst*dt); agg: "mean" or "max" (also
    torch.mean / torch.max: `agg` of mnist_node.jl:69,:80,:98 -- `maximum` for stiff_est).  With "max" the saved values' cotangent is lambda at the
    largest one, which the host has to pick: the step then runs as three library calls instead of one.
    sync=False: the reverse pass is only enqueued (rnde_node_backward_async) and the losses come back as device tensors, so a
    training loop can queue the optimiser update and the next step underneath it; nothing is read on the host.
    flat (dataparallel.FlatGrads over model.trainable()): the reverse pass writes both gradients straight into that one
    contiguous buffer; reducer (dataparallel.GradientAllReducer on the same buffer): ONE sum-all-reduce of the whole buffer
    behind the reverse pass; averaging is left to the optimiser (reducer.grad_scale)."""
    import ctypes as C
    from . import _lib
    node = model.node
    L = _lib.lib()
    from .node import _check_f32
    for name, t in (("x", x), ("y", y), ("p2", model.p2), ("p3", model.p3)):
        _check_f32(name, t)
    x2 = x.reshape(x.shape[0], -1).contiguous()
    B, D = x2.shape
    node._func = node.resolve_func(func)
    agg_max = agg in ("max", "maximum", torch.max)
    if not agg_max and agg not in ("mean", torch.mean):
        raise ValueError("agg: 'mean' or 'max' (mnist_node.jl:69,:80,:98)")
    h = node._acquire(x2, True)
    ts = node.tspan if tspan is None else [float(tspan[0]), float(tspan[1])]
    stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
    if reducer is not None:
        # the all-reduce below works on the reducer's OWN flat buffer: the gradients have to be written into its views, or the
        # collective would sum a buffer nothing wrote and leave p.grad un-reduced (replicas would silently diverge)
        if flat is None:
            flat = reducer.fg
        elif flat.flat.data_ptr() != reducer.fg.flat.data_ptr():
            raise ValueError("fused_loss_and_grad: `flat` is not the buffer `reducer` reduces (pass reducer.fg, or leave flat=None)")
    if flat is not None:                                  # trainable() = (p1 (empty), p2, p3) -> groups [p2, p3]
        p2bar, p3bar = flat.views[0], flat.views[1]
    else:
        p2bar, p3bar = torch.empty_like(model.p2), torch.empty_like(model.p3)
    if not sync and not agg_max and node.col_tile == 0 and (reducer is None or reducer.comm is not None) and os.environ.get("RNDE_ONE_CALL", "1") != "0":   # (0: A/B switch)
        # the whole step gradient as ONE library call (rnde_node_classifier_grad): head and reverse-sweep packs are queued before
        # the forward's host wait, so the GPU does not idle between the solve and its reverse
        ce = torch.empty(1, dtype=torch.float32, device=x2.device)
        nfe, reg_h = C.c_int64(0), C.c_float(0.0)
        _lib.check(h.ptr, L.rnde_node_classifier_grad(
            h.ptr, x2.data_ptr(), model.p2.detach().data_ptr(), model.p3.detach().data_ptr(), y.contiguous().data_ptr(), B,
            model.post_shape[1], ts[0], ts[1], float(lam) if regularize else 0.0, p2bar.data_ptr(), p3bar.data_ptr(), None,
            ce.data_ptr(), C.byref(reg_h), C.byref(nfe), reducer.comm if reducer is not None else None, stream))
        model.p2.grad, model.p3.grad = p2bar, p3bar
        node.last_nfe = int(nfe.value)
        reg = float(reg_h.value)
        return ce + reg, ce, reg, int(nfe.value)
    u = torch.empty_like(x2)
    nfe, nsv = C.c_int64(0), C.c_int32(0)
    sv = (C.c_float * (node.max_attempts + 1))()
    p2, p3 = model.p2.detach(), model.p3.detach()
    _lib.check(h.ptr, L.rnde_node_forward(h.ptr, x2.data_ptr(), p2.data_ptr(), B, ts[0], ts[1], u.data_ptr(), C.byref(nfe), sv,
                                          C.byref(nsv), 1, stream))
    n_cls = model.post_shape[1]
    ubar = torch.empty_like(x2)
    ce = torch.empty(1, dtype=torch.float32, device=x2.device)
    _lib.check(h.ptr, L.rnde_classifier_head(h.ptr, u.data_ptr(), p3.data_ptr(), y.contiguous().data_ptr(), B, n_cls, None,
                                             ubar.data_ptr(), p3bar.data_ptr(), ce.data_ptr(), stream))
    n = nsv.value
    reg = 0.0
    svb = None
    if regularize and n > 0 and node.regularize:
        if agg_max:                                       # lambda * maximum(sv.saveval): the cotangent sits on the (first) largest value
            vals = list(sv[:n])
            k = max(range(n), key=lambda i: vals[i])
            reg = lam * vals[k]
            svb = (C.c_float * n)(*[lam if i == k else 0.0 for i in range(n)])
        else:
            reg = lam * sum(sv[:n]) / n                   # lambda * mean(sv.saveval)
            svb = (C.c_float * n)(*([lam / n] * n))
    xbar = torch.empty_like(x2)
    n2 = p2bar.numel()
    if not sync:
        _lib.check(h.ptr, L.rnde_node_backward_async(h.ptr, ubar.data_ptr(), svb, xbar.data_ptr(), p2bar.data_ptr(), None, stream))
        if reducer is not None:
            reducer.allreduce_range_(0, n2 + p3bar.numel())
        model.p2.grad, model.p3.grad = p2bar, p3bar
        node.last_nfe = int(nfe.value)
        model._keep = (ubar, xbar, u)            # buffers the enqueued kernels still use
        return ce + reg, ce, reg, int(nfe.value)
    _lib.check(h.ptr, L.rnde_node_backward(h.ptr, ubar.data_ptr(), svb, xbar.data_ptr(), p2bar.data_ptr(), None, stream))
    if reducer is not None:
        reducer.allreduce_range_(0, n2 + p3bar.numel())
    model.p2.grad, model.p3.grad = p2bar, p3bar
    node.last_nfe = int(nfe.value)
    ce_f = float(ce.item())
    return ce_f + reg, ce_f, reg, int(nfe.value)


def logitcrossentropy(pred, y_onehot):
    """Flux.Losses.logitcrossentropy: mean over the batch of -sum(y .* logsoftmax(pred))."""
    return -(y_onehot * torch.log_softmax(pred, dim=1)).sum(dim=1).mean()


def loss_function(x, y, model, p1=None, p2=None, p3=None, lam=1.0e2, regularize=True, agg=torch.mean, func="error_est",
                  tspan=None):
    """experiments/mnist_node.jl:132-137: cross entropy + lambda * agg(sv.saveval)."""
    pred, nfe, sv = model(x, p1, p2, p3, func=func, tspan=tspan)
    ce = logitcrossentropy(pred, y)
    reg = lam * agg(sv.saveval) if regularize else torch.zeros((), device=pred.device)
    return ce + reg, ce.detach(), reg.detach(), nfe


class FluxOptimiser:
    """Flux.Optimise.Optimiser(InvDecay(gamma), Momentum(eta, rho)) applied group by group
    (reference src/utils.jl:149-156 skips empty groups)."""

    def __init__(self, params, gamma=1.0e-5, eta=0.1, rho=0.9):
        self.params = [p for p in params if p.numel() > 0]
        self.gamma, self.eta, self.rho = gamma, eta, rho
        self.n = [1 for _ in self.params]                 # InvDecay state starts at 1
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self, grads=None, grad_scale=1.0):
        """grad_scale: factor applied to the gradients first (1 / world after a sum-all-reduce: the averaging rides in this launch)."""
        for i, p in enumerate(self.params):
            g = p.grad if grads is None else grads[i]
            if g is None:
                continue
            if p.is_cuda:                                   # one launch per group through the C ABI (rnde_momentum_step)
                import ctypes as C
                from . import _lib
                g = g.contiguous()
                st = _lib.lib().rnde_momentum_step_scaled(p.data_ptr(), g.data_ptr(), self.v[i].data_ptr(), p.numel(), self.n[i],
                                                          self.gamma, self.eta, self.rho, float(grad_scale),
                                                          C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
                _lib.check(None, st)
                self.n[i] += 1
                p.grad = None
                continue
            g = g * (grad_scale / (1.0 + self.gamma * self.n[i]))   # InvDecay (host tensors: the same recurrence in torch ops)
            self.n[i] += 1
            self.v[i].mul_(self.rho).sub_(g, alpha=self.eta)   # Momentum: v = rho v - eta g ; x -= -v
            p.add_(self.v[i])
            p.grad = None


class FluxADAM:
    """Flux.Optimise.ADAM(eta, (beta1, beta2)), optionally behind InvDecay (experiments/mnist_nsde.jl: Optimiser(InvDecay(1.0e-5), ADAM(0.01))), group by group, one launch per group through
    the C ABI (rnde_adam_step); same recurrence as torch.optim.Adam (tests/test_gpu_layer.py)."""

    def __init__(self, params, eta=0.001, beta=(0.9, 0.999), eps=1.0e-8, gamma=0.0):
        """gamma > 0: Flux.Optimise.Optimiser(InvDecay(gamma), ADAM(eta)) as experiments/mnist_nsde.jl builds it -- InvDecay scales the gradient
        by 1 / (1 + gamma n) (n = 1, 2, ... per group) BEFORE ADAM sees it; the factor rides in the launch's gradient scale."""
        self.params = [p for p in params if p.numel() > 0]
        self.eta, self.beta, self.eps, self.gamma = eta, beta, eps, gamma
        self.n = [1 for _ in self.params]                 # InvDecay state starts at 1
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        import ctypes as C
        from . import _lib
        self.t += 1
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                continue
            if not p.is_cuda:
                raise RuntimeError("FluxADAM runs on the device only")
            g = g.contiguous()
            st = _lib.lib().rnde_adam_step(p.data_ptr(), g.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr(), p.numel(), self.t, self.eta,
                                           self.beta[0], self.beta[1], self.eps, float(grad_scale) / (1.0 + self.gamma * self.n[i]),
                                           C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
            _lib.check(None, st)
            self.n[i] += 1
            p.grad = None


def sample_tspan_ubound(b=0.5, generator=None):
    """STEER (experiments/mnist_node.jl:104-105,:133): integrate to t1 ~ U(1 - b, 1 + b) instead of 1."""
    r = float(torch.rand((), generator=generator))
    return [0.0, 1.0 - (2.0 * r - 1.0) * b]


# (lambda0, lambda1, callback, aggregation, solver) per regulariser type, experiments/mnist_node.jl:62-103
REGULARISERS = {
    "error_est": (1.0e2, 1.0e1, "error_est", torch.mean, "Tsit5"),
    "stiff_est": (0.1, 0.1, "stiff_est", torch.max, "AutoTsit5"),
    "error_stiff_est": (1.0e1, 1.0e1, "error_stiff_est", torch.mean, "AutoTsit5"),
}


def lambda_schedule(epoch, epochs=75, lam0=1.0e2, lam1=1.0e1):
    """experiments/mnist_node.jl:65-66,:106-108: exponential decay from lam0 to lam1 over the run."""
    k = math.log(lam0 / lam1) / epochs
    return lam0 * math.exp(-k * epoch)


def accuracy(model, batches):
    """reference src/metrics.jl:4-18 (argmax match), without the forced GC."""
    correct = total = 0
    with torch.no_grad():
        for x, y in batches:
            pred, _, _ = model(x)
            correct += (pred.argmax(1) == y.argmax(1)).sum().item()
            total += x.shape[0]
    return correct / max(total, 1)
