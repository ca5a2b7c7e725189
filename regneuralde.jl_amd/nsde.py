"""TrackedNeuralDSDE and ClassifierNSDE: the reference's stochastic layer (src/models/neural_sde.jl) and its caller
(src/models/supervised_classification.jl:50-103), same constructors and call contracts, with the `solve` call replaced by
librnde.so (rnde_nsde_*: the whole adaptive SOSRI solve is one kernel launch).

    nsde = TrackedNeuralDSDE(Chain(Dense(32, 64, "tanh"), Dense(64, 32)), Chain(Dense(32, 32)), [0.0, 1.0], regularize, "SOSRI",
                             save_everystep=False, reltol=1.4e-1, abstol=1.4e-1, save_start=False)     # experiments/mnist_nsde.jl:72-84
    u, nfe1, nfe2, sv = nsde(x, p)          # x: (B, D) cuda tensor == Julia D x B

With `saveat=` the {R,true} methods run (neural_sde.jl:44-61,:84-113; experiments/sde_toy_problem.jl:50-60): `u` is then the
(B, T, D) tensor whose memory is exactly the Julia D x T x B array of diffeqsol_to_3dtrackedarray (src/utils.jl:17-19).

`func`: the caller's closure `(u, t, integrator) -> value` as the experiment passes it (`model(x, p1, p2, p3, p4; func = save_func)`, mnist_nsde.jl) --
recognised as one of the reference's two SDE callbacks (node.py::reg_code: EEst*dt, mnist_nsde.jl:48, or |eigen_est| / 10.6 with the composite
solver AutoSOSRI2(SOSRI2()), :51-61 -- what the shipped configs/mnist_nsde.yml selects), never called per step -- or its name ('error_est' /
'stiff_est').  Differences inherent to the host language / the device:
`save_everystep=True` (a result whose length is data dependent): `rnde_nsde_forward_everystep`, the state after every accepted step;
the noise comes from the library's Philox stream (seed = nsde.seed, advanced every call) unless `noise=` passes a pool of
standard normals of shape (n_pool, 2, B, D) -- a Julia caller would fill that from its own RNG.
"""
import os
import ctypes as C

import torch

from . import _lib
from .layers import Chain, Dense, destructure
from .node import MAX_HANDLES_PER_KEY, SOSRI2_STABILITY_SIZE, SavedValues, _check_f32, _TapeToken, effective_reg, reg_code

_ACT = {"identity": 0, "tanh": 1}


class _NsdeHandle:
    def __init__(self, cfg):
        self.ptr = C.c_void_p()
        _lib.check_nsde(None, _lib.lib().rnde_nsde_create(C.byref(cfg), C.byref(self.ptr)))
        self.busy = False

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().rnde_nsde_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _SdeSolve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, layer, keep_tape, noise, seed, saveat):
        h = layer._acquire(x)
        L = _lib.lib()
        B, D = x.shape
        n1, n2, nsv = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        sv_host = (C.c_float * (layer.max_attempts + 1))()
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        nptr, npool = (noise.data_ptr() if noise is not None else None), (0 if noise is None else noise.shape[0])
        if saveat is None:
            u = torch.empty_like(x)
            st = L.rnde_nsde_forward(h.ptr, x.data_ptr(), p.data_ptr(), B, layer.tspan[0], layer.tspan[1], nptr, npool, seed,
                                     u.data_ptr(), C.byref(n1), C.byref(n2), sv_host, C.byref(nsv), 1 if keep_tape else 0, stream)
        elif saveat == "everystep":      # save_everystep = true (neural_sde.jl:14): the state after every accepted step, count known after the call
            cap = layer.max_attempts + 1
            buf = torch.empty((B, cap, D), dtype=torch.float32, device=x.device)      # (filled as (B, n, D) from its start, see node.py)
            th, nout = (C.c_float * cap)(), C.c_int32(0)
            st = L.rnde_nsde_forward_everystep(h.ptr, x.data_ptr(), p.data_ptr(), B, layer.tspan[0], layer.tspan[1], nptr, npool, seed,
                                               1 if layer.kwargs.get("save_start", True) else 0, buf.data_ptr(), cap, th, C.byref(nout),
                                               C.byref(n1), C.byref(n2), sv_host, C.byref(nsv), 1 if keep_tape else 0, stream)
            _lib.check_nsde(h.ptr, st)
            n = nout.value
            u = buf.reshape(-1)[: B * n * D].reshape(B, n, D).clone()
            layer.last_times = [float(th[i]) for i in range(n)]
        else:
            T = len(saveat)
            u = torch.empty((B, T, D), dtype=torch.float32, device=x.device)
            sa = (C.c_float * T)(*saveat)
            st = L.rnde_nsde_forward_saveat(h.ptr, x.data_ptr(), p.data_ptr(), B, layer.tspan[0], layer.tspan[1], nptr, npool, seed, sa, T,
                                            u.data_ptr(), C.byref(n1), C.byref(n2), sv_host, C.byref(nsv), 1 if keep_tape else 0, stream)
        _lib.check_nsde(h.ptr, st)
        layer.last_nfe = (int(n1.value), int(n2.value))
        saveval = torch.tensor(list(sv_host[:nsv.value]), dtype=torch.float32, device=x.device)
        ctx.layer, ctx.h, ctx.nsv = layer, h, nsv.value
        if keep_tape:
            h.busy = True
            ctx.token = _TapeToken(h, lambda ptr: None)     # the SDE tape needs no release call: freeing the handle for reuse is enough
        return u, saveval

    @staticmethod
    def backward(ctx, u_bar, sv_bar):
        layer, h = ctx.layer, ctx.h
        u_bar = u_bar.contiguous().to(torch.float32)
        x_bar = torch.empty((u_bar.shape[0], u_bar.shape[-1]), dtype=torch.float32, device=u_bar.device)
        p_bar = torch.empty(layer.P, dtype=torch.float32, device=u_bar.device)
        svb = None
        if ctx.nsv and sv_bar is not None:
            svb = (C.c_float * ctx.nsv)(*sv_bar.detach().to("cpu", torch.float32).tolist())
        stream = C.c_void_p(torch.cuda.current_stream(u_bar.device).cuda_stream)
        st = _lib.lib().rnde_nsde_backward(h.ptr, u_bar.data_ptr(), svb, x_bar.data_ptr(), p_bar.data_ptr(), stream)
        ctx.token.finish()
        _lib.check_nsde(h.ptr, st)
        return x_bar, p_bar, None, None, None, None, None


class TrackedNeuralDSDE:
    """Mirror of reference src/models/neural_sde.jl:1-41 (struct + constructor) and :64-82,:116-146 (the {R,false} call methods)."""

    def __init__(self, model1, model2, tspan, regularize, solver="SOSRI", *, max_batch=512, max_attempts=256, cb_save_start=True,
                 seed=0, **kwargs):
        if solver not in _lib.SDE_SOLVER:
            raise ValueError("solver: SOSRI (experiments/mnist_nsde.jl:49,:63), SOSRI2, AutoSOSRI2 (= AutoSOSRI2(SOSRI2()), :60) or SRIW1")
        if isinstance(model2, Dense):
            model2 = Chain(model2)
        self.save_everystep = bool(kwargs.get("save_everystep", False)) and "saveat" not in kwargs      # (saveat given: it decides what is saved)
        self.return_multiple = bool(kwargs.get("save_everystep", False)) or "saveat" in kwargs          # neural_sde.jl:14
        self.last_times = None
        if model1.time_dep or model2.time_dep:
            raise ValueError("drift and diffusion are time independent (neural_sde.jl:45-52 call re(p)(u))")
        self.model1, self.model2 = model1, model2
        p1, p2 = destructure(model1), destructure(model2)
        self.p = torch.cat([p1, p2])                       # neural_sde.jl:17
        self.len = p1.numel()                              # neural_sde.jl:38
        self.P = self.p.numel()
        self.tspan = [float(tspan[0]), float(tspan[1])]
        self.regularize = bool(regularize)
        self.solver, self.kwargs = solver, dict(kwargs)
        self.max_batch, self.max_attempts, self.cb_save_start = int(max_batch), int(max_attempts), bool(cb_save_start)
        self.seed = int(seed)
        self._handles = {}
        self.last_nfe = None
        self._reg = 1            # rnde_reg code of the current call's callback (set by __call__)

    def _config(self, device_index):
        cfg = _lib.NsdeConfig()
        for name, model in (("drift", self.model1), ("diff", self.model2)):
            dims = model.dims()
            setattr(cfg, f"{name}_layers", len(model.layers))
            for i, d in enumerate(dims):
                getattr(cfg, f"{name}_dims")[i] = d
            for i, l in enumerate(model.layers):
                getattr(cfg, f"{name}_act")[i] = _ACT[l.act]
        cfg.max_batch = self.max_batch
        cfg.solver = _lib.SDE_SOLVER[self.solver]
        cfg.reltol = float(self.kwargs.get("reltol", 1e-2))   # StochasticDiffEq defaults when not given
        cfg.abstol = float(self.kwargs.get("abstol", 1e-2))
        cfg.regularize = self._reg if self.regularize else 0
        cfg.cb_save_start = int(self.cb_save_start)
        cfg.max_attempts = self.max_attempts
        cfg.device = device_index
        return cfg

    def _acquire(self, x):
        key = x.device.index or 0
        hs = self._handles.setdefault((key, self._reg if self.regularize else 0), [])
        for h in hs:
            if not h.busy:
                return h
        if len(hs) >= MAX_HANDLES_PER_KEY:
            raise RuntimeError(f"{len(hs)} taped forwards of this layer are pending without a backward pass")
        h = _NsdeHandle(self._config(key))
        hs.append(h)
        return h

    def reg_of(self, func):
        """rnde_reg code of a call's `func`: None = the layer's default callback EEst*dt (neural_sde.jl:87); a name; or the caller's closure, recognised as
        the reference's EEst*dt / |eigen_est| / 10.6 (mnist_nsde.jl:48, :53-58).  The stiffness estimate exists for the composite AutoSOSRI2(SOSRI2()) only."""
        if not self.regularize:
            return 0
        if callable(func):
            code = effective_reg(reg_code(func, SOSRI2_STABILITY_SIZE), self.solver == "AutoSOSRI2")
        elif func in (None, "error_est"):
            code = 1
        elif func == "stiff_est":
            code = 2
        else:
            raise ValueError("func: a callback (u, t, integrator) -> value, or 'error_est' (EEst*dt, neural_sde.jl:87) / 'stiff_est' "
                             "(|eigen_est| / 10.6, experiments/mnist_nsde.jl:51-61)")
        if code in (3, 4):
            raise ValueError("the SDE layer records EEst*dt or the stiffness estimate (mnist_nsde.jl:45-61), not their blend and not |eigen_est*dt|")
        if code == 2 and self.solver != "AutoSOSRI2":      # names and closures alike (effective_reg checked the closure above)
            raise ValueError("the stiffness estimate of an SRI step is filled by the composite AutoSOSRI2(SOSRI2()) only (mnist_nsde.jl:60); a plain "
                             "solver -- SOSRI2 included -- leaves integrator.eigen_est at its initial value")
        return code

    def __call__(self, x, p=None, func=None, noise=None):
        """(x, p = n.p; func) -> (arr, nfe1, nfe2, sv)   [neural_sde.jl:64-82, :116-146]"""
        if not x.is_cuda:
            raise RuntimeError("TrackedNeuralDSDE runs on the MI355X only: x must be a cuda tensor (no CPU fallback)")
        self._reg = self.reg_of(func)
        p = self.p if p is None else p
        if p.device != x.device:
            if p is self.p:
                self.p = p = self.p.to(x.device)
            else:
                raise RuntimeError("p and x must live on the same device")
        _check_f32("p", p)
        _check_f32("x", x)
        if noise is not None:
            _check_f32("noise", noise)
            if tuple(noise.shape[1:]) != (2, x.shape[0], x.shape[1]) or not noise.is_cuda:
                raise ValueError("noise: cuda tensor of shape (n_pool, 2, B, D)")
            noise = noise.contiguous()
        keep = torch.is_grad_enabled() and (x.requires_grad or p.requires_grad)
        self.seed += 1
        times = None
        if self.return_multiple and self.save_everystep:
            times = "everystep"
        elif self.return_multiple:
            from .node import TrackedNeuralODE
            times = TrackedNeuralODE._saveat_times(self.kwargs["saveat"], self.tspan)
        u, saveval = _SdeSolve.apply(x.contiguous(), p.contiguous(), self, keep, noise, self.seed, times)
        nfe1, nfe2 = self.last_nfe                         # n.nfes, reset after the solve (neural_sde.jl:78-79,:142-143)
        return u, nfe1, nfe2, (SavedValues(saveval) if self.regularize else None)


def _expand(x, d):
    """supervised_classification.jl:102-103: repeat along the batch dimension (Julia's last dimension = torch's first)."""
    return x.repeat(d, *([1] * (x.dim() - 1)))


class ClassifierNSDE:
    """Mirror of reference src/models/supervised_classification.jl:50-100: presde -> nsde -> postsde, `trajectories` sample
    paths per input, logits averaged over them."""

    def __init__(self, presde, nsde, postsde, device=None):
        self.nsde = nsde
        self.pre_shape, self.post_shape = (presde.n_in, presde.n_out), (postsde.n_in, postsde.n_out)
        dev = device if device is not None else torch.device("cuda", 0)
        self.p1 = destructure(Chain(presde)).to(dev).requires_grad_(True)
        self.p2 = nsde.p.to(dev).requires_grad_(True)
        self.p3 = destructure(Chain(postsde)).to(dev).requires_grad_(True)

    def trainable(self):                                   # Flux.trainable(m::ClassifierNSDE) = (m.p1, m.p2, m.p3)
        return (self.p1, self.p2, self.p3)

    @staticmethod
    def _dense(x, p, shape):
        n_in, n_out = shape
        return x @ p[: n_in * n_out].view(n_in, n_out) + p[n_in * n_out:]

    def __call__(self, x, p1=None, p2=None, p3=None, trajectories=10, **nsde_kwargs):
        p1 = self.p1 if p1 is None else p1
        p2 = self.p2 if p2 is None else p2
        p3 = self.p3 if p3 is None else p3
        bsize = x.shape[0]
        x = _expand(x.reshape(bsize, -1), trajectories)                           # :92 (batch order: trajectory-major, as repeat gives)
        h = self._dense(x, p1, self.pre_shape)                                   # :93-94
        u, nfe1, nfe2, sv = self.nsde(h.contiguous(), p2, **nsde_kwargs)         # :95
        z = self._dense(u, p3, self.post_shape)                                  # :96-97
        z = z.reshape(trajectories, bsize, -1).mean(dim=0)                       # :98
        return z, nfe1, nfe2, sv


def nsde_loss_function(x, y, model, p1=None, p2=None, p3=None, trajectories=1, lam=1.0e2, regularize=True, agg=torch.mean, func="error_est"):
    """experiments/mnist_nsde.jl:88-118 (without the logger): logitcrossentropy(pred, y) + lambda * agg(sv.saveval).  `func`: the experiment's
    `save_func` (a closure, or 'error_est' / 'stiff_est': mnist_nsde.jl:45-61)."""
    from .classifier import logitcrossentropy
    pred, nfe1, nfe2, sv = model(x, p1, p2, p3, trajectories=trajectories, func=func if regularize else None)
    ce = logitcrossentropy(pred, y)
    reg = lam * agg(sv.saveval) if (regularize and sv is not None) else torch.zeros((), device=pred.device)
    return ce + reg, ce, reg, nfe1, nfe2


def fused_nsde_loss_and_grad(model, x, y, trajectories=1, lam=1.0e2, regularize=True, func="error_est"):
    """One training-step gradient of `nsde_loss_function` (agg = mean) without a tape library in the loop, the counterpart of
    classifier.fused_loss_and_grad for ClassifierNSDE: [Dense pre-layer] -> [solve, taped: ONE launch] -> [Dense post-layer, trajectory
    mean, logitcrossentropy and their reverse as a handful of matrix products] -> [reverse sweep: ONE launch] -> [pre-layer gradient].
    Sets .grad on p1, p2, p3; returns (total_loss, cross_entropy, reg, nfe1, nfe2) with the losses as device tensors / floats.
    Same arithmetic as autograd through ClassifierNSDE.__call__ (tests/test_gpu_nsde.py compares the two)."""
    nsde = model.nsde
    nsde._reg = nsde.reg_of(func)                    # which callback the handle records (mnist_nsde.jl:45-61): EEst*dt or the stiffness estimate
    for name, t in (("x", x), ("y", y), ("p1", model.p1), ("p2", model.p2), ("p3", model.p3)):
        _check_f32(name, t)
    L = _lib.lib()
    with torch.no_grad():
        bsize = x.shape[0]
        xe = _expand(x.reshape(bsize, -1), trajectories).contiguous()
        p1, p2, p3 = model.p1.detach(), model.p2.detach().contiguous(), model.p3.detach()
        n_in, n_h = model.pre_shape
        W1, b1 = p1[: n_in * n_h].view(n_in, n_h), p1[n_in * n_h:]
        h = torch.addmm(b1, xe, W1)                                              # supervised_classification.jl:93-94
        B, D = h.shape
        hd = nsde._acquire(h)
        n_u, n_c = model.post_shape
        stream = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        if trajectories == 1 and n_c <= 16 and os.environ.get("RNDE_ONE_CALL", "1") != "0":
            # solve + postsde/loss + their reverse + reverse sweep as ONE library call (rnde_nsde_classifier_grad): the head is queued before the
            # forward's host wait, the GPU does not idle between the solve and its reverse  (RNDE_ONE_CALL=0: the three calls below, for A/B)
            n1, n2, reg_h = C.c_int64(0), C.c_int64(0), C.c_float(0.0)
            p3c = p3.contiguous()
            hbar, p2bar, p3bar = torch.empty_like(h), torch.empty_like(p2), torch.empty_like(p3c)
            ce_t = torch.empty(1, dtype=torch.float32, device=h.device)
            nsde.seed += 1
            _lib.check_nsde(hd.ptr, L.rnde_nsde_classifier_grad(
                hd.ptr, h.data_ptr(), p2.data_ptr(), p3c.data_ptr(), y.contiguous().data_ptr(), B, n_c, nsde.tspan[0], nsde.tspan[1], None, 0,
                nsde.seed, float(lam) if (regularize and nsde.regularize) else 0.0, p2bar.data_ptr(), p3bar.data_ptr(), hbar.data_ptr(),
                ce_t.data_ptr(), C.byref(reg_h), C.byref(n1), C.byref(n2), stream))
            nsde.last_nfe = (int(n1.value), int(n2.value))
            p1bar = torch.cat([(xe.t() @ hbar).reshape(-1), hbar.sum(dim=0)])
            model._keep = (hbar, h, p3c)          # buffers the enqueued kernels still use
            model.p1.grad, model.p2.grad, model.p3.grad = p1bar, p2bar, p3bar
            reg = float(reg_h.value)
            return ce_t[0] + reg, ce_t[0], reg, int(n1.value), int(n2.value)
        n1, n2, nsv = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        sv_host = (C.c_float * (nsde.max_attempts + 1))()
        stream = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        u = torch.empty_like(h)
        nsde.seed += 1
        _lib.check_nsde(hd.ptr, L.rnde_nsde_forward(hd.ptr, h.data_ptr(), p2.data_ptr(), B, nsde.tspan[0], nsde.tspan[1], None, 0, nsde.seed,
                                                     u.data_ptr(), C.byref(n1), C.byref(n2), sv_host, C.byref(nsv), 1, stream))
        nsde.last_nfe = (int(n1.value), int(n2.value))
        n_u, n_c = model.post_shape
        W3, b3 = p3[: n_u * n_c].view(n_u, n_c), p3[n_u * n_c:]
        if trajectories == 1 and n_c <= 16:     # postsde + loss + their reverse: three launches through the C ABI (rnde_nsde_classifier_head)
            ubar, p3bar = torch.empty_like(u), torch.empty_like(p3)
            ce_t = torch.empty(1, dtype=torch.float32, device=u.device)
            _lib.check_nsde(hd.ptr, L.rnde_nsde_classifier_head(hd.ptr, u.data_ptr(), p3.contiguous().data_ptr(), y.contiguous().data_ptr(), B, n_c, None,
                                                                ubar.data_ptr(), p3bar.data_ptr(), ce_t.data_ptr(), stream))
            ce = ce_t[0]
        else:
            z = torch.addmm(b3, u, W3).view(trajectories, bsize, n_c).mean(dim=0)    # :96-98
            logp = torch.log_softmax(z, dim=1)
            ce = -(y * logp).sum() / bsize                                           # Flux.Losses.logitcrossentropy
            dz = ((torch.exp(logp) * y.sum(dim=1, keepdim=True) - y) / (bsize * trajectories)).repeat(trajectories, 1)
            p3bar = torch.cat([(u.t() @ dz).reshape(-1), dz.sum(dim=0)])
            ubar = (dz @ W3.t()).contiguous()
        n = nsv.value
        reg, svb = 0.0, None
        if regularize and nsde.regularize and n > 0:
            reg = lam * sum(sv_host[:n]) / n                                     # lambda * mean(sv.saveval)
            svb = (C.c_float * n)(*([lam / n] * n))
        hbar = torch.empty_like(h)
        p2bar = torch.empty_like(p2)
        # (asynchronous form: the pre-layer gradient and the optimiser update queue up behind the reverse sweep, nothing waits on the host)
        _lib.check_nsde(hd.ptr, L.rnde_nsde_backward_async(hd.ptr, ubar.data_ptr(), svb, hbar.data_ptr(), p2bar.data_ptr(), stream))
        p1bar = torch.cat([(xe.t() @ hbar).reshape(-1), hbar.sum(dim=0)])
        model._keep = (ubar, hbar, u, h)          # buffers the enqueued kernels still use
        model.p1.grad, model.p2.grad, model.p3.grad = p1bar, p2bar, p3bar
    return ce + reg, ce, reg, int(n1.value), int(n2.value)
