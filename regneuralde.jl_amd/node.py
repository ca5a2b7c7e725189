"""TrackedNeuralODE: the reference's DE layer (src/models/neural_ode.jl), same constructor and
call contract, with the `solve` call replaced by librnde.so (hand-written gfx950 kernels).

    node = TrackedNeuralODE(model, [0.0, 1.0], True, regularize, "Tsit5",
                            save_everystep=False, reltol=1.4e-8, abstol=1.4e-8, save_start=False)
    u, nfe, sv = node(x, p)            # x: (B, D) cuda tensor == Julia D x B;  sv.saveval: tensor or None

With `saveat=` (constructor keyword or per-call override, neural_ode.jl:35-46) the {R,true} methods run
(neural_ode.jl:79-108,:146-180): `u` is then the (B, T, D) tensor whose memory is exactly the Julia
D x T x B array of diffeqsol_to_3dtrackedarray (src/utils.jl:17-19).

`func` is the caller's closure `(u, t, integrator) -> value` as the reference passes it (mnist_node.jl:134) -- it is evaluated on two
mock integrators and RECOGNISED as one of the reference's callbacks (mnist_node.jl:67,:74-79,:88-97; `reg_code` below), the library
computes the value inside its kernels -- or that callback's name ('error_est' / 'stiff_est' / 'error_stiff_est').  Anything else raises.
Differences from the Julia layer that are inherent to the host language:
`save_everystep=True` (a result whose length is data dependent; no reference call site uses it): `rnde_node_forward_everystep`, the state
after every accepted step (and the initial one with save_start=True), (B, n, D) with n known after the call.
"""
import ctypes as C

import torch

from . import _lib
from .layers import destructure

_ACT = {"identity": 0, "tanh": 1}
_FUNCS = {None: 1, "none": 0, "error_est": 1, "stiff_est": 2, "error_stiff_est": 3, "stiff_est_dt": 4}      # None: the layer's default callback (neural_ode.jl:116)
_FUNC_NAMES = {0: "none", 1: "error_est", 2: "stiff_est", 3: "error_stiff_est", 4: "stiff_est_dt"}
TSIT5_STABILITY_SIZE = 3.5068     # OrdinaryDiffEq.alg_stability_size(Tsit5()): what mnist_node.jl:73,:86 divides by
SOSRI2_STABILITY_SIZE = 10.6      # StochasticDiffEq.alg_stability_size(SOSRI2()): mnist_nsde.jl:55


class MockIntegrator:
    """What a saving callback reads off the integrator (neural_ode.jl:116; mnist_node.jl:67,:76,:89-91)."""

    def __init__(self, EEst, dt, eigen_est):
        self.EEst, self.dt, self.eigen_est = EEst, dt, eigen_est


_PROBES = ((2.0, 3.0, 5.0), (0.5, 0.25, -7.0))


def reg_code(func, stability_size):
    """rnde_reg code (include/rnde.h) of a caller's callback `func(u, t, integrator)`: its values on two mock integrators are matched against
    the reference's callbacks -- 0 (neural_ode.jl:54), EEst*dt (mnist_node.jl:67), |eigen_est|/stability_size (:74-79), EEst*dt +
    0.1*eigen_est/stability_size (:88-97), |eigen_est*dt| (test/test_node.jl:75).  The same rule as bindings/julia/RNDE.jl::reg_code.  Raises ValueError for anything else: a
    regulariser the kernels do not compute must not be replaced by another one silently."""
    got = [float(func(None, 0.0, MockIntegrator(*m))) for m in _PROBES]
    s = float(stability_size)
    want = {0: lambda e, d, g: 0.0, 1: lambda e, d, g: e * d, 2: lambda e, d, g: abs(g) / s, 3: lambda e, d, g: e * d + 0.1 * g / s,
            4: lambda e, d, g: abs(g * d)}      # 4: the reference's own test, func = abs(integrator.eigen_est * integrator.dt) (test/test_node.jl:75,:84)
    for code in (0, 1, 2, 3, 4):
        if all(abs(v - want[code](*m)) <= 1e-4 * abs(want[code](*m)) + 1e-7 for v, m in zip(got, _PROBES)):
            return code
    raise ValueError("func is none of the callbacks librnde.so computes (EEst*dt, |eigen_est|/stability_size, EEst*dt + "
                     f"0.1*eigen_est/stability_size, |eigen_est*dt|, 0): on (EEst, dt, eigen_est) = {_PROBES[0]} and {_PROBES[1]} it returned {got}")


def effective_reg(code, composite):
    """The callback code a (func, solver) pair runs with, or an error: integrator.eigen_est is filled by the composite solvers only
    (AutoTsit5(Tsit5()), AutoSOSRI2(SOSRI2()): mnist_node.jl:81,:99, mnist_nsde.jl:60).  Under a plain solver it keeps its INITIAL value -- 1, not 0
    [RECALL] -- so the reference would record a constant there (|1| / stability_size, EEst*dt + 0.1 / stability_size), which no experiment does and the
    kernels do not reproduce: every callback that reads eigen_est is refused under a plain solver, the blend included (round 5 mapped the blend to
    EEst*dt, i.e. assumed eigen_est = 0: a constant offset of the saved values)."""
    if composite or code in (0, 1):
        return code
    raise ValueError("func reads integrator.eigen_est, which only the composite solvers (AutoTsit5 / AutoSOSRI2) fill; with a plain solver "
                     "the reference records its initial value, a constant -- build the layer with the composite solver")


class SavedValues:
    """DiffEqCallbacks.SavedValues as returned by the regularised call (neural_ode.jl:126,:143)."""

    def __init__(self, saveval):
        self.saveval = saveval


class _Handle:
    def __init__(self, cfg):
        self.ptr = C.c_void_p()
        st = _lib.lib().rnde_node_create(C.byref(cfg), C.byref(self.ptr))
        _lib.check(None, st)
        self.busy = False

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().rnde_node_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _TapeToken:
    """Lifetime of one taped forward.  It lives in the autograd node's ctx: when the graph is dropped WITHOUT a backward pass
    (the reference's per-epoch `_, nfe, _ = node(dummy)` with tracking on, experiments/mnist_node.jl:236) the token dies with
    it, releases the tape and hands the handle back to the pool -- otherwise every such call would pin a handle and its
    max_attempts arena (~4 GB at B = 512) for good."""

    def __init__(self, h, release):
        self.h, self.release, self.done = h, release, False

    def finish(self):
        if not self.done:
            self.done = True
            self.h.busy = False

    def __del__(self):
        try:
            if not self.done:
                self.done = True
                if self.h.ptr:
                    self.release(self.h.ptr)
                self.h.busy = False
        except Exception:
            pass


MAX_HANDLES_PER_KEY = 4   # simultaneously pending (taped, not yet back-propagated) forwards per layer, device and callback


def _check_f32(name, t):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype}): the C ABI reads raw fp32 memory; convert explicitly, "
                        "as the reference does with eltype(p)")


class _Solve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, layer, t0, t1, keep_tape, saveat):
        h = layer._acquire(x, keep_tape)
        L = _lib.lib()
        B, D = x.shape
        nfe = C.c_int64(0)
        nsv = C.c_int32(0)
        sv_host = (C.c_float * (layer.max_attempts + 1))()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if saveat is None:
            u = torch.empty_like(x)
            st = L.rnde_node_forward(h.ptr, x.data_ptr(), p.data_ptr(), B, t0, t1, u.data_ptr(), C.byref(nfe), sv_host,
                                     C.byref(nsv), 1 if keep_tape else 0, C.c_void_p(stream))
        elif saveat == "everystep":      # save_everystep = true: every accepted step's end (neural_ode.jl:10-11); the count comes back with the call
            cap = layer.max_attempts + 1
            buf = torch.empty((B, cap, D), dtype=torch.float32, device=x.device)
            # the library writes D x n x B column-major = (B, n, D) rows of the n it finds: ask with the capacity as the time stride is n, not cap --
            # so the call goes into a scratch of the full capacity first and the (B, n, D) block is what it filled
            th = (C.c_float * cap)()
            nout = C.c_int32(0)
            st = L.rnde_node_forward_everystep(h.ptr, x.data_ptr(), p.data_ptr(), B, t0, t1, 1 if layer.kwargs.get("save_start", True) else 0,
                                               buf.data_ptr(), cap, th, C.byref(nout), C.byref(nfe), sv_host, C.byref(nsv), 1 if keep_tape else 0,
                                               C.c_void_p(stream))
            _lib.check(h.ptr, st)
            n = nout.value
            u = buf.reshape(-1)[: B * n * D].reshape(B, n, D).clone()
            layer.last_times = [float(th[i]) for i in range(n)]
        else:
            T = len(saveat)
            u = torch.empty((B, T, D), dtype=torch.float32, device=x.device)
            sa = (C.c_float * T)(*saveat)
            st = L.rnde_node_forward_saveat(h.ptr, x.data_ptr(), p.data_ptr(), B, t0, t1, sa, T, u.data_ptr(), C.byref(nfe),
                                            sv_host, C.byref(nsv), 1 if keep_tape else 0, C.c_void_p(stream))
        _lib.check(h.ptr, st)
        layer.last_nfe = int(nfe.value)
        saveval = torch.tensor(list(sv_host[:nsv.value]), dtype=torch.float32, device=x.device)
        ctx.layer, ctx.h, ctx.nsv = layer, h, nsv.value
        if keep_tape:
            h.busy = True
            ctx.token = _TapeToken(h, L.rnde_node_release_tape)
        return u, saveval

    @staticmethod
    def backward(ctx, u_bar, sv_bar):
        layer, h = ctx.layer, ctx.h
        L = _lib.lib()
        u_bar = u_bar.contiguous().to(torch.float32)
        x_bar = torch.empty((u_bar.shape[0], u_bar.shape[-1]), dtype=torch.float32, device=u_bar.device)
        p_bar = torch.empty(layer.P, dtype=torch.float32, device=u_bar.device)
        svb = None
        if ctx.nsv and sv_bar is not None:
            svb = (C.c_float * ctx.nsv)(*sv_bar.detach().to("cpu", torch.float32).tolist())
        tsb = (C.c_float * 2)()
        stream = torch.cuda.current_stream(u_bar.device).cuda_stream
        st = L.rnde_node_backward(h.ptr, u_bar.data_ptr(), svb, x_bar.data_ptr(), p_bar.data_ptr(), tsb, C.c_void_p(stream))
        ctx.token.finish()
        _lib.check(h.ptr, st)
        layer.last_tspan_bar = (tsb[0], tsb[1])
        return x_bar, p_bar, None, None, None, None, None


class TrackedNeuralODE:
    """Mirror of reference src/models/neural_ode.jl:1-33 (struct + constructor) and :48-180 (call methods)."""

    def __init__(self, model, tspan, time_dep, regularize, solver="Tsit5", *, max_batch=512, max_attempts=128,
                 cb_save_start=True, track_ctrl=True, track_initdt=True, col_tile=0, matrix_mode=None, **kwargs):
        if solver not in ("Tsit5", "AutoTsit5", "DP5", "DOP853"):
            raise ValueError("solver: the reference's call sites use Tsit5() / AutoTsit5(Tsit5()) only; DP5 (a second 7-stage pair) and DOP853 "
                             "(a 13-stage table) run on the tableau-as-data kernels (Dense chains of width <= 64)")
        self.solver = solver
        self.model = model
        self.p = destructure(model)                      # Flux.destructure (neural_ode.jl:12)
        self.tspan = [float(tspan[0]), float(tspan[1])]
        self.time_dep = bool(time_dep)
        self.regularize = bool(regularize)
        self.kwargs = dict(kwargs)                       # reltol, abstol, save_everystep, save_start, saveat
        self.return_multiple = bool(kwargs.get("save_everystep", False)) or ("saveat" in kwargs)  # neural_ode.jl:11
        self.save_everystep = bool(kwargs.get("save_everystep", False)) and "saveat" not in kwargs      # (saveat given: it decides what is saved)
        if bool(time_dep) != bool(model.time_dep):
            raise ValueError("time_dep must match the model (TDChain => True)")
        self.max_batch, self.max_attempts = int(max_batch), int(max_attempts)
        self.cb_save_start, self.track_ctrl, self.track_initdt, self.col_tile = cb_save_start, track_ctrl, track_initdt, col_tile
        # which unit forms the Dense-layer products of the one-launch solve (include/rnde.h: rnde_node_set_matrix_mode): None = the library's default
        # (bf16x3 on the matrix cores where the kernels serve the shape, RNDE_X3 overrides), 0 = fp32-input MFMA, 1 = bf16x3
        self.matrix_mode = matrix_mode
        self.P = self.p.numel()
        self._handles = {}
        self._coupling = None
        self.last_nfe = None
        self.last_times = None                            # save_everystep: the times of the states the last call returned
        self.last_tspan_bar = None

    # -- C-ABI config -------------------------------------------------------------------------
    def _config(self, device_index, func):
        cfg = _lib.NodeConfig()
        dims = self.model.dims()
        cfg.n_layers = len(self.model.layers)
        for i, d in enumerate(dims):
            cfg.dims[i] = d
        for i, l in enumerate(self.model.layers):
            cfg.act[i] = _ACT[l.act]
        cfg.time_dep = int(self.time_dep)
        cfg.pre_act = int(getattr(self.model, "pre_act", False))
        cfg.max_batch = self.max_batch
        cfg.solver = _lib.ODE_SOLVER[self.solver]
        cfg.reltol = float(self.kwargs.get("reltol", 1e-3))   # OrdinaryDiffEq defaults when not given
        cfg.abstol = float(self.kwargs.get("abstol", 1e-6))
        cfg.regularize = _FUNCS[func] if self.regularize else 0
        cfg.cb_save_start = int(self.cb_save_start)
        cfg.track_ctrl = int(self.track_ctrl)
        cfg.track_initdt = int(self.track_initdt)
        cfg.max_attempts = self.max_attempts
        cfg.device = device_index
        cfg.col_tile = self.col_tile
        return cfg

    def _acquire(self, x, keep_tape, func=None):
        key = (x.device.index or 0, bool(keep_tape), self._func)
        hs = self._handles.setdefault(key, [])
        for h in hs:
            if not h.busy:
                return h
        if len(hs) >= MAX_HANDLES_PER_KEY:
            raise RuntimeError(f"{len(hs)} taped forwards of this layer are pending without a backward pass; each owns a tape of "
                               "max_attempts records.  Run them under torch.no_grad() (NFE probes), call backward, or drop the graphs")
        h = _Handle(self._config(x.device.index or 0, self._func))
        if self.matrix_mode is not None:
            _lib.check(h.ptr, _lib.lib().rnde_node_set_matrix_mode(h.ptr, int(self.matrix_mode)))
        if self._coupling is not None:
            _lib.check(h.ptr, _lib.lib().rnde_node_set_coupling(h.ptr, self._coupling[0], self._coupling[1]))
        hs.append(h)
        return h

    def set_coupling(self, comm, global_batch):
        """SURVEY 8e mode 2 (include/rnde.h: rnde_node_set_coupling): ONE step-size controller for all shards of a minibatch split by
        columns over `world` layers.  `comm`: an rnde_comm* (ctypes void pointer: `GradientAllReducer.comm` across processes,
        rnde_comm_create_local_group inside one); None switches back to independent controllers.  Every rank must make the same calls."""
        self._coupling = None if comm is None else (comm, int(global_batch))
        for hs in self._handles.values():
            for h in hs:
                _lib.check(h.ptr, _lib.lib().rnde_node_set_coupling(h.ptr, comm, int(global_batch) if comm is not None else 0))

    @staticmethod
    def _saveat_times(saveat, ts):
        """What OrdinaryDiffEq saves for `saveat`: a vector as given (must lie in [t0, t1], increasing); a number s is the
        range t0:s:t1 plus t1 (save_start = save_end = true for a Number)."""
        if isinstance(saveat, (int, float)):
            s, out, k = float(saveat), [], 0
            if s <= 0:
                raise ValueError("saveat step must be positive")
            while ts[0] + k * s < ts[1]:
                out.append(ts[0] + k * s)
                k += 1
            out.append(ts[1])
            return out
        v = [float(t) for t in (saveat.detach().cpu().reshape(-1).tolist() if torch.is_tensor(saveat) else list(saveat))]
        if not v:
            raise ValueError("saveat is empty")
        if any(b <= a for a, b in zip(v, v[1:])) or v[0] < ts[0] or v[-1] > ts[1]:
            raise ValueError("saveat must be strictly increasing inside tspan")
        return v

    def resolve_func(self, func):
        """The name of the library callback a call's `func` selects (None when the layer does not regularise): a closure `(u, t, integrator) -> value`
        as the reference passes it (mnist_node.jl:134) is recognised, not called per step; names pass through."""
        if not self.regularize:
            return None
        if callable(func):
            func = _FUNC_NAMES[reg_code(func, TSIT5_STABILITY_SIZE)]
        if func not in _FUNCS:
            raise ValueError("func must be a callback (u, t, integrator) -> value or one of None/'error_est', 'stiff_est', "
                             "'error_stiff_est' (the three callbacks of experiments/mnist_node.jl:62-103), 'stiff_est_dt' (test/test_node.jl:75)")
        effective_reg(_FUNCS[func], self.solver == "AutoTsit5")      # names and closures alike: a callback that reads eigen_est needs the composite solver
        return func

    # -- call operator ------------------------------------------------------------------------
    def __call__(self, x, p=None, func=None, tspan=None, saveat=None):
        """(x, p = n.p; func, tspan, saveat) -> (res, nfe, sv)   [neural_ode.jl:48-54,:76,:110-119,:143]"""
        if saveat is not None and not self.return_multiple:
            raise ValueError("saveat override on a layer built without saveat: the reference dispatches on the constructor "
                             "keyword (neural_ode.jl:11), build the layer with saveat=")
        if not x.is_cuda:
            raise RuntimeError("TrackedNeuralODE runs on the MI355X only: x must be a cuda tensor (no CPU fallback)")
        p = self.p if p is None else p
        if p.device != x.device:
            if p is self.p:
                self.p = p = self.p.to(x.device)
            else:
                raise RuntimeError("p and x must live on the same device")
        _check_f32("p", p)
        _check_f32("x", x)
        x2 = x.reshape(x.shape[0], -1).contiguous()
        ts = self.tspan if tspan is None else [float(tspan[0]), float(tspan[1])]   # _convert_tspan, utils.jl:21-23
        self._func = self.resolve_func(func)
        keep = torch.is_grad_enabled() and (x2.requires_grad or p.requires_grad)
        times = None
        if self.return_multiple and self.save_everystep and saveat is None:
            times = "everystep"
        elif self.return_multiple:      # update_saveat! (neural_ode.jl:35-46): a per-call override, the stored one otherwise
            times = self._saveat_times(self.kwargs["saveat"] if saveat is None else saveat, ts)
        u, saveval = _Solve.apply(x2, p.contiguous(), self, ts[0], ts[1], keep, times)
        sv = SavedValues(saveval) if self.regularize else None
        return u, self.last_nfe, sv
