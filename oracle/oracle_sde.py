"""ctypes loader for the CPU oracle of the stochastic path (TEST INFRASTRUCTURE ONLY -- see rnde_sde_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Parity versus the Julia
reference is UNPINNED (rnde_sde_oracle.h says what pins it instead).
"""
import ctypes as C

import numpy as np

from .oracle import Arch, _cap_threads, build, make_arch


def arch_nsde_drift(D=32, H=64):
    """experiments/mnist_nsde.jl:73: Chain(Dense(32, 64, tanh), Dense(64, 32)), time independent."""
    return make_arch([D, H, D], ["tanh", "identity"], False)


def arch_nsde_diffusion(D=32):
    """experiments/mnist_nsde.jl:74: Dense(32, 32)."""
    return make_arch([D, D], ["identity"], False)


TABLEAU = {"SOSRI": 0, "SRIW1": 1, "SOSRI2": 2}


class SriTableau(C.Structure):
    _fields_ = [(n, C.c_double * 16) for n in ("A0", "A1", "B0", "B1")] + \
               [(n, C.c_double * 4) for n in ("alpha", "beta1", "beta2", "beta3", "beta4", "c0", "c1")] + \
               [("order", C.c_double), ("delta", C.c_double)]


def sri_tableau(name, dtype=np.float64):
    lib = C.CDLL(build()[1])
    T = SriTableau()
    assert lib.orc_sri_tableau_get(C.c_int(TABLEAU[name]), C.byref(T)) == 0
    out = {n: np.array(getattr(T, n)).reshape(4, 4) for n in ("A0", "A1", "B0", "B1")}
    out.update({n: np.array(getattr(T, n)) for n in ("alpha", "beta1", "beta2", "beta3", "beta4", "c0", "c1")})
    out["order"], out["delta"] = T.order, T.delta
    return out


class SdeOracle:
    def __init__(self, drift, diffusion, dtype=np.float32, reltol=0.14, abstol=0.14, tableau="SOSRI", reg_kind=1,
                 cb_save_start=1, max_attempts=2048, **ctrl):
        libs = build()
        self.dtype = np.dtype(dtype)
        f64 = self.dtype == np.float64
        self.lib = C.CDLL(libs[1] if f64 else libs[0])
        _cap_threads(self.lib, libs[1] if f64 else libs[0])
        self.real = C.c_double if f64 else C.c_float

        class Config(C.Structure):
            _fields_ = [("drift", Arch), ("diffusion", Arch), ("reltol", self.real), ("abstol", self.real),
                        ("tableau", C.c_int), ("reg_kind", C.c_int), ("cb_save_start", C.c_int), ("max_attempts", C.c_int)] + \
                       [(n, self.real) for n in ("beta1", "beta2", "gamma", "qmin", "qmax", "qoldinit", "delta", "stability_size")]

        self.cfg = Config(drift, diffusion, reltol, abstol, TABLEAU[tableau], reg_kind, cb_save_start, max_attempts,
                          *[ctrl.get(n, 0.0) for n in ("beta1", "beta2", "gamma", "qmin", "qmax", "qoldinit", "delta", "stability_size")])
        L = self.lib
        L.orc_sde_create.restype = C.c_void_p
        L.orc_sde_destroy.argtypes = [C.c_void_p]
        L.orc_sde_param_count.restype = C.c_int
        self.D = drift.dims[0]
        ld = C.c_int(0)
        self.P = L.orc_sde_param_count(C.byref(self.cfg), C.byref(ld))
        self.len = ld.value
        self.h = C.c_void_p(L.orc_sde_create(C.byref(self.cfg)))
        assert self.h
        self.max_attempts = max_attempts

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_sde_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def attempt(self, p, uprev, dt, dW, dZ):
        p, uprev, dW, dZ = map(self._arr, (p, uprev, dW, dZ))
        B = uprev.shape[0]
        kg = np.empty((8, B, self.D), dtype=self.dtype)
        unew = np.empty_like(uprev)
        e = self.real(0)
        self.lib.orc_sde_attempt(self.h, self._p(p), self._p(uprev), C.c_int(B), self.real(dt), self._p(dW), self._p(dZ),
                                 self._p(kg), self._p(unew), C.byref(e))
        return kg, unew, float(e.value)

    def set_replay(self, dt=None, acc=None):
        if dt is None:
            self.lib.orc_sde_set_replay(self.h, None, None, C.c_int(0))
            return
        d = self._arr(dt); a = np.ascontiguousarray(acc, dtype=np.int32)
        self.lib.orc_sde_set_replay(self.h, self._p(d), a.ctypes.data_as(C.c_void_p), C.c_int(len(d)))

    def set_saveat(self, saveat=None):
        if saveat is None:
            self.lib.orc_sde_set_saveat(self.h, None, C.c_int(0))
            self.nsave = 0
            return
        sa = self._arr(saveat)
        self.lib.orc_sde_set_saveat(self.h, self._p(sa), C.c_int(len(sa)))
        self.nsave = len(sa)

    def forward(self, x, p, noise, t0=0.0, t1=1.0):
        """x: (B, D); noise: (n_pool, 2, B, D) standard normals.  After set_saveat: u is (B, T, D)."""
        x, p, noise = self._arr(x), self._arr(p), self._arr(noise)
        B = x.shape[0]
        assert noise.shape[1:] == (2, B, self.D)
        ns = getattr(self, "nsave", 0)
        u = np.empty((B, ns, self.D), dtype=self.dtype) if ns else np.empty_like(x)
        n1, n2 = C.c_long(0), C.c_long(0)
        sv = np.zeros(self.max_attempts + 1, dtype=self.dtype)
        nsv, natt, ndr = C.c_int(0), C.c_int(0), C.c_int(0)
        log = np.zeros((self.max_attempts, 4), dtype=self.dtype)
        rc = self.lib.orc_sde_forward(self.h, self._p(x), self._p(p), C.c_int(B), self.real(t0), self.real(t1), self._p(noise),
                                      C.c_int(noise.shape[0]), self._p(u), C.byref(n1), C.byref(n2), self._p(sv), C.byref(nsv),
                                      self._p(log), C.byref(natt), C.byref(ndr))
        return dict(rc=rc, u=u, nfe1=n1.value, nfe2=n2.value, saveval=sv[:nsv.value].copy(), steps=log[:natt.value].copy(),
                    nattempts=natt.value, ndraws=ndr.value)

    def eigen_norms(self, n_max=None):
        """(n_acc, 2): rms(k4 - k3), rms(H0_4 - H0_3) of every accepted step of the last forward (eigen_est = their quotient); after `attempt`: (1, 2)."""
        out = np.zeros((n_max or self.max_attempts + 1, 2), dtype=self.dtype)
        n = self.lib.orc_sde_eigen_norms(self.h, self._p(out))
        return out[:max(n, 1)].copy()

    def path_total(self, B):
        w = np.empty((B, self.D), dtype=self.dtype); z = np.empty_like(w)
        self.lib.orc_sde_path_total(self.h, self._p(w), self._p(z))
        return w, z

    def backward(self, ubar, svbar=None):
        ubar = self._arr(ubar)
        xbar = np.empty((ubar.shape[0], self.D), dtype=self.dtype)
        pbar = np.empty(self.P, dtype=self.dtype)
        sv = self._arr(svbar) if svbar is not None else None
        rc = self.lib.orc_sde_backward(self.h, self._p(ubar), self._p(sv) if sv is not None else None, self._p(xbar), self._p(pbar))
        assert rc == 0, rc
        return xbar, pbar


def nsde_params(drift, diffusion, rng, dtype=np.float32, scale=1.0, diff_scale=1.0):
    """[p_drift; p_diffusion] in Flux.destructure order (neural_sde.jl:15-17), Glorot-uniform weights, zero biases."""
    from .oracle import glorot_params
    return np.concatenate([glorot_params(drift, rng, dtype, scale), glorot_params(diffusion, rng, dtype, diff_scale)]).astype(dtype)
