"""Shared helpers for the parity tests: device-side wrappers that go through the C ABI only."""
import ctypes as C

import numpy as np
import torch

import regneuralde_jl_amd as rn
from regneuralde_jl_amd import _lib
from oracle.oracle import Oracle, arch_mnist, arch_test_node, glorot_params, make_arch  # test infrastructure


def make_cfg(dims, acts, max_batch, reltol=1.4e-8, abstol=1.4e-8, regularize=1, max_attempts=128, col_tile=0,
             cb_save_start=1, track_ctrl=1, track_initdt=1, time_dep=1, pre_act=0, persist=0, wgrad_side_pct=0, stage_generic=0,
             solver="Tsit5"):
    cfg = _lib.NodeConfig()
    cfg.n_layers = len(acts)
    for i, d in enumerate(dims):
        cfg.dims[i] = d
    for i, a in enumerate(acts):
        cfg.act[i] = {"identity": 0, "tanh": 1}[a]
    cfg.time_dep = int(time_dep)
    cfg.pre_act = int(pre_act)
    cfg.max_batch = max_batch
    cfg.solver = _lib.ODE_SOLVER[solver]
    cfg.reltol, cfg.abstol = reltol, abstol
    cfg.regularize = regularize
    cfg.cb_save_start, cfg.track_ctrl, cfg.track_initdt = cb_save_start, track_ctrl, track_initdt
    cfg.max_attempts = max_attempts
    cfg.device = 0
    cfg.col_tile = col_tile
    cfg.persist, cfg.wgrad_side_pct, cfg.stage_generic = persist, wgrad_side_pct, stage_generic
    return cfg


class Node:
    """Thin RAII wrapper over the C ABI handle (device pointers in, device pointers out)."""

    def __init__(self, cfg, matrix_mode=None):
        """matrix_mode: None = the library's default (include/rnde.h: rnde_node_set_matrix_mode), 0 = fp32-input MFMA, 1 = bf16x3 on the matrix cores"""
        self.L = _lib.lib()
        self.h = C.c_void_p()
        _lib.check(None, self.L.rnde_node_create(C.byref(cfg), C.byref(self.h)))
        if matrix_mode is not None:
            _lib.check(self.h, self.L.rnde_node_set_matrix_mode(self.h, int(matrix_mode)))
        self.cfg = cfg
        self.D = cfg.dims[0]
        self.stream = None            # HIP stream of forward / backward (None: the default stream); see own_stream()
        self._tstream = None

    def own_stream(self):
        """Give this handle a stream of its own (two handles driven from two host threads must not share the default stream)."""
        self._tstream = torch.cuda.Stream()
        self.stream = C.c_void_p(self._tstream.cuda_stream)
        return self

    def close(self):
        if self.h:
            self.L.rnde_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def feval(self, u, p, t):
        ud, pd = self.dev(u), self.dev(p)
        out = torch.empty_like(ud)
        _lib.check(self.h, self.L.rnde_debug_feval(self.h, ud.data_ptr(), pd.data_ptr(), u.shape[0], t, out.data_ptr(), None))
        return out.cpu().numpy()

    def attempt(self, uprev, k1, p, t, dt):
        B = uprev.shape[0]
        ud, kd, pd = self.dev(uprev), self.dev(k1), self.dev(p)
        nk = 12 if self.cfg.solver == _lib.ODE_SOLVER["DOP853"] else 6      # stages after k1: 6 for the 7-stage pairs, S - 1 for a table
        kout = torch.empty((nk, B, self.D), dtype=torch.float32, device="cuda")
        unew = torch.empty_like(ud)
        eest = C.c_float(0)
        _lib.check(self.h, self.L.rnde_debug_attempt(self.h, ud.data_ptr(), kd.data_ptr(), pd.data_ptr(), B, t, dt,
                                                     kout.data_ptr(), unew.data_ptr(), C.byref(eest), None))
        return kout.cpu().numpy(), unew.cpu().numpy(), eest.value

    def forward(self, x, p, t0=0.0, t1=1.0, keep_tape=False):
        B = x.shape[0]
        xd, pd = self.dev(x), self.dev(p)
        u = torch.empty_like(xd)
        nfe = C.c_int64(0)
        nsv = C.c_int32(0)
        sv = (C.c_float * (self.cfg.max_attempts + 1))()
        # inputs were staged on torch's stream; self.stream may be another one (and a device-wide wait could wait on a peer handle
        # that is waiting for this one)
        torch.cuda.current_stream().synchronize()
        st = self.L.rnde_node_forward(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, u.data_ptr(), C.byref(nfe), sv,
                                      C.byref(nsv), int(keep_tape), self.stream)
        _lib.check(self.h, st)
        steps = (C.c_float * (4 * self.cfg.max_attempts))()
        natt = C.c_int32(0)
        self.L.rnde_node_steps(self.h, steps, self.cfg.max_attempts, C.byref(natt))
        return dict(u=u.cpu().numpy(), nfe=nfe.value, saveval=np.array(sv[:nsv.value], dtype=np.float32),
                    steps=np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4), nattempts=natt.value)

    def forward_replay(self, x, p, dtp, acc, t0=0.0, t1=1.0, keep_tape=False):
        """The solve along a given sequence of (proposed step size, accept decision) pairs: rnde_node_forward_replay."""
        B = x.shape[0]
        xd, pd = self.dev(x), self.dev(p)
        u = torch.empty_like(xd)
        nfe = C.c_int64(0)
        nsv = C.c_int32(0)
        sv = (C.c_float * (self.cfg.max_attempts + 1))()
        n = len(dtp)
        pairs = (C.c_float * (2 * n))()
        for i in range(n):
            pairs[2 * i], pairs[2 * i + 1] = float(dtp[i]), float(acc[i] != 0)
        st = self.L.rnde_node_forward_replay(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, pairs, n, u.data_ptr(), C.byref(nfe), sv,
                                             C.byref(nsv), int(keep_tape), None)
        _lib.check(self.h, st)
        steps = (C.c_float * (4 * self.cfg.max_attempts))()
        natt = C.c_int32(0)
        self.L.rnde_node_steps(self.h, steps, self.cfg.max_attempts, C.byref(natt))
        return dict(u=u.cpu().numpy(), nfe=nfe.value, saveval=np.array(sv[:nsv.value], dtype=np.float32),
                    steps=np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4), nattempts=natt.value)

    def forward_saveat(self, x, p, saveat, t0=0.0, t1=1.0, keep_tape=False):
        B = x.shape[0]
        xd, pd = self.dev(x), self.dev(p)
        T = len(saveat)
        u = torch.empty((B, T, self.D), dtype=torch.float32, device="cuda")
        nfe = C.c_int64(0)
        nsv = C.c_int32(0)
        sv = (C.c_float * (self.cfg.max_attempts + 1))()
        sa = (C.c_float * T)(*[float(v) for v in saveat])
        st = self.L.rnde_node_forward_saveat(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, sa, T, u.data_ptr(), C.byref(nfe), sv,
                                             C.byref(nsv), int(keep_tape), None)
        _lib.check(self.h, st)
        steps = (C.c_float * (4 * self.cfg.max_attempts))()
        natt = C.c_int32(0)
        self.L.rnde_node_steps(self.h, steps, self.cfg.max_attempts, C.byref(natt))
        return dict(u=u.cpu().numpy(), nfe=nfe.value, saveval=np.array(sv[:nsv.value], dtype=np.float32),
                    steps=np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4), nattempts=natt.value)

    def backward(self, ubar, svbar=None):
        B = ubar.shape[0]
        ub = self.dev(ubar)
        xb = torch.empty((B, self.D), dtype=torch.float32, device="cuda")   # ubar is (B, T, D) after a saveat forward
        P = self.L.rnde_param_count(C.byref(self.cfg))
        pb = torch.empty(P, dtype=torch.float32, device="cuda")
        tsb = (C.c_float * 2)()
        svb = None if svbar is None else (C.c_float * len(svbar))(*[float(v) for v in svbar])
        torch.cuda.current_stream().synchronize()
        st = self.L.rnde_node_backward(self.h, ub.data_ptr(), svb, xb.data_ptr(), pb.data_ptr(), tsb, self.stream)
        _lib.check(self.h, st)
        return xb.cpu().numpy(), pb.cpu().numpy(), np.array([tsb[0], tsb[1]], dtype=np.float32)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ---- stochastic layer (rnde_nsde_*) ------------------------------------------------------------------------------------
def make_nsde_cfg(drift_dims, drift_acts, diff_dims, diff_acts, max_batch, reltol=0.14, abstol=0.14, solver="SOSRI", regularize=1,
                  cb_save_start=1, max_attempts=256, generic=0, **ctrl):
    cfg = _lib.NsdeConfig()
    cfg.drift_layers = len(drift_acts)
    for i, d in enumerate(drift_dims):
        cfg.drift_dims[i] = d
    for i, a in enumerate(drift_acts):
        cfg.drift_act[i] = {"identity": 0, "tanh": 1}[a]
    cfg.diff_layers = len(diff_acts)
    for i, d in enumerate(diff_dims):
        cfg.diff_dims[i] = d
    for i, a in enumerate(diff_acts):
        cfg.diff_act[i] = {"identity": 0, "tanh": 1}[a]
    cfg.max_batch = max_batch
    cfg.solver = _lib.SDE_SOLVER[solver]
    cfg.reltol, cfg.abstol = reltol, abstol
    cfg.regularize, cfg.cb_save_start, cfg.max_attempts, cfg.device = regularize, cb_save_start, max_attempts, 0
    for k in ("beta1", "beta2", "gamma", "qmin", "qmax", "qoldinit", "delta", "stability_size"):
        setattr(cfg, k, ctrl.get(k, 0.0))
    cfg.generic = generic
    return cfg


class NsdeNode:
    """Thin wrapper over the C ABI handle of the stochastic layer (device pointers in, device pointers out)."""

    def __init__(self, cfg):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        _lib.check_nsde(None, self.L.rnde_nsde_create(C.byref(cfg), C.byref(self.h)))
        self.cfg = cfg
        self.D = cfg.drift_dims[0]
        ld = C.c_int32(0)
        self.P = self.L.rnde_nsde_param_count(C.byref(cfg), C.byref(ld))
        self.len = ld.value

    def close(self):
        if self.h:
            self.L.rnde_nsde_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    dev = staticmethod(Node.dev)

    def attempt(self, uprev, p, dt, dW, dZ):
        B = uprev.shape[0]
        ud, pd, wd, zd = self.dev(uprev), self.dev(p), self.dev(dW), self.dev(dZ)
        kg = torch.empty((8, B, self.D), dtype=torch.float32, device="cuda")
        un = torch.empty_like(ud)
        e = C.c_float(0)
        _lib.check_nsde(self.h, self.L.rnde_nsde_debug_attempt(self.h, ud.data_ptr(), pd.data_ptr(), B, dt, wd.data_ptr(), zd.data_ptr(),
                                                               kg.data_ptr(), un.data_ptr(), C.byref(e), None))
        return kg.cpu().numpy(), un.cpu().numpy(), e.value

    def _log(self):
        steps = (C.c_float * (4 * self.cfg.max_attempts))()
        natt, ndr = C.c_int32(0), C.c_int32(0)
        self.L.rnde_nsde_steps(self.h, steps, self.cfg.max_attempts, C.byref(natt), C.byref(ndr))
        return np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4), natt.value, ndr.value

    def forward(self, x, p, noise=None, seed=0, t0=0.0, t1=1.0, keep_tape=False, replay=None, check=True, saveat=None):
        """noise: (n_pool, 2, B, D) standard normals or None (library stream from `seed`); replay: (n, 2) array of (dt, accepted);
        saveat: times -> u is (B, T, D)."""
        B = x.shape[0]
        xd, pd = self.dev(x), self.dev(p)
        nd = None if noise is None else self.dev(noise)
        u = torch.empty_like(xd) if saveat is None else torch.empty((B, len(saveat), self.D), dtype=torch.float32, device="cuda")
        n1, n2, nsv = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        sv = (C.c_float * (self.cfg.max_attempts + 1))()
        npool = 0 if noise is None else noise.shape[0]
        if saveat is not None:
            sa = (C.c_float * len(saveat))(*[float(v) for v in saveat])
            st = self.L.rnde_nsde_forward_saveat(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, nd.data_ptr() if nd is not None else None, npool,
                                                 seed, sa, len(saveat), u.data_ptr(), C.byref(n1), C.byref(n2), sv, C.byref(nsv), int(keep_tape), None)
        elif replay is None:
            st = self.L.rnde_nsde_forward(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, nd.data_ptr() if nd is not None else None, npool,
                                          seed, u.data_ptr(), C.byref(n1), C.byref(n2), sv, C.byref(nsv), int(keep_tape), None)
        else:
            rp = np.ascontiguousarray(replay, dtype=np.float32)
            st = self.L.rnde_nsde_forward_replay(self.h, xd.data_ptr(), pd.data_ptr(), B, t0, t1, nd.data_ptr(), npool,
                                                 rp.ctypes.data_as(C.POINTER(C.c_float)), rp.shape[0], u.data_ptr(), C.byref(n1), C.byref(n2), sv,
                                                 C.byref(nsv), int(keep_tape), None)
        if check:
            _lib.check_nsde(self.h, st)
        steps, natt, ndr = self._log()
        return dict(rc=st, u=u.cpu().numpy(), nfe1=n1.value, nfe2=n2.value, saveval=np.array(sv[:nsv.value], dtype=np.float32), steps=steps,
                    nattempts=natt, ndraws=ndr)

    def backward(self, ubar, svbar=None):
        ub = self.dev(ubar)
        xb = torch.empty((ub.shape[0], self.D), dtype=torch.float32, device="cuda")   # ubar is (B, T, D) after a saveat forward
        pb = torch.empty(self.P, dtype=torch.float32, device="cuda")
        svb = None if svbar is None else (C.c_float * len(svbar))(*[float(v) for v in svbar])
        _lib.check_nsde(self.h, self.L.rnde_nsde_backward(self.h, ub.data_ptr(), svb, xb.data_ptr(), pb.data_ptr(), None))
        return xb.cpu().numpy(), pb.cpu().numpy()
