"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY -- see rnde_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package never does.  Parity versus the Julia reference is UNPINNED
(no Julia runtime and no golden vectors exist; rnde_oracle.h explains what pins it instead).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LAYERS = 8


def build(force=False):
    out = os.path.join(_HERE, "_build")
    libs = [os.path.join(out, f"librnde_oracle_{s}.so") for s in ("f32", "f64")]
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("rnde_oracle.c", "rnde_oracle.h", "rnde_sde_oracle.c", "rnde_sde_oracle.h"))
    if force or not all(os.path.exists(l) and os.path.getmtime(l) >= src_m for l in libs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return libs


def effective_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota.  The GPU boxes show 256 hardware threads
    under a quota of 16 CPUs; OpenMP's default (one thread per visible CPU) then oversubscribes 16x and a barrier-heavy loop nest runs
    ~300x slower than on 16 threads (gpurun_out/r03/cpu_threads.log: 2.0 against 721 samples/s)."""
    n = len(os.sched_getaffinity(0))
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                     # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):      # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
    except Exception:
        pass
    return n


_threads_set = set()


def _cap_threads(lib, path):
    """Once per loaded library: OpenMP threads = effective_cores() unless OMP_NUM_THREADS says otherwise."""
    if path in _threads_set:
        return
    _threads_set.add(path)
    if "OMP_NUM_THREADS" not in os.environ:
        try:
            lib.orc_set_threads(C.c_int(effective_cores()))
        except AttributeError:
            pass


class Arch(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dims", C.c_int * (MAX_LAYERS + 1)), ("act", C.c_int * MAX_LAYERS),
                ("time_dep", C.c_int), ("pre_act", C.c_int)]


def make_arch(dims, acts, time_dep, pre_act=False):
    a = Arch()
    a.n_layers = len(acts)
    for i, d in enumerate(dims):
        a.dims[i] = d
    for i, x in enumerate(acts):
        a.act[i] = {"identity": 0, "tanh": 1, 0: 0, 1: 1}[x]
    a.time_dep = int(time_dep)
    a.pre_act = int(pre_act)
    return a


# the three dynamics shapes the reference ships
def arch_mnist(D=784, H=100):
    """experiments/mnist_node.jl:41-54: Dense(D+1,H,tanh) -> Dense(H+1,D,tanh), t appended to both inputs."""
    return make_arch([D, H, D], ["tanh", "tanh"], True)


def arch_test_node():
    """test/test_node.jl:4: TDChain(Dense(3,10,tanh), Dense(11,2))."""
    return make_arch([2, 10, 2], ["tanh", "identity"], True)


def arch_latent():
    """experiments/latent_ode.jl:113-124: tanh, then 8 x Dense(20<->50, tanh), time independent."""
    return make_arch([20, 50, 20, 50, 20, 50, 20, 50, 20], ["tanh"] * 8, False, pre_act=True)


class Oracle:
    def __init__(self, arch, dtype=np.float32, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1, cb_save_start=1,
                 track_ctrl=1, track_initdt=1, max_attempts=4096, solver="Tsit5", sum_order=0):
        """sum_order (fp32 only; rnde_oracle.c `orc_set_sum_order`): 0 = sequential-k dot products and libm tanh (a textbook CPU);
        bit 0 = the two-layer TDChain's GEMMs accumulated in the device stage engine's order (split-K row blocks, two interleaved
        accumulators of K = 4 FMA chains); bit 1 = tanh by the device's formula.  3 = "what the device computes in matrix mode 0" (the fp32-input MFMA), to
        rounding of v_exp_f32 / v_rcp_f32.  Bit 2 (two-layer TDChain only): the GEMMs as the device's matrix mode 1 forms them (csrc/rnde_x3.h: operands split
        exactly into three bf16 numbers, six cross products, every 32-term matrix instruction as four exact 8-term sums added with a rounding each; the two f of the
        initial-step rule in the fp32-MFMA order, as the device evaluates them) -- 7 = "what the device computes by default" since round 6.
        The mode is process-wide in the C library; this wrapper sets it before every call it makes."""
        libs = build()
        self.dtype = np.dtype(dtype)
        f64 = self.dtype == np.float64
        self.lib = C.CDLL(libs[1] if f64 else libs[0])
        _cap_threads(self.lib, libs[1] if f64 else libs[0])
        self.real = C.c_double if f64 else C.c_float
        self.sum_order = int(sum_order)

        class Config(C.Structure):
            _fields_ = [("arch", Arch), ("reltol", self.real), ("abstol", self.real), ("reg_kind", C.c_int),
                        ("cb_save_start", C.c_int), ("track_ctrl", C.c_int), ("track_initdt", C.c_int),
                        ("max_attempts", C.c_int), ("solver", C.c_int)]

        self.Config = Config
        self.solver = {"Tsit5": 0, "DP5": 1, "DOP853": 2}[solver]
        self.lib.orc_stage_count.restype = C.c_int
        self.S = int(self.lib.orc_stage_count(C.c_int(self.solver)))      # stages incl. the closing (first-same-as-last) one: 7, 7, 13
        self.cfg = Config(arch, reltol, abstol, reg_kind, cb_save_start, track_ctrl, track_initdt, max_attempts, self.solver)
        self.arch = arch
        L = self.lib
        L.orc_param_count.restype = C.c_int
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_initdt.restype = self.real
        self.D = arch.dims[0]
        self.P = L.orc_param_count(C.byref(arch))
        self.h = C.c_void_p(L.orc_create(C.byref(self.cfg)))
        self.max_attempts = max_attempts

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    # x: (B, D) row-major numpy == D x B column-major Julia
    def f_eval(self, p, u, t):
        self.lib.orc_set_sum_order(C.c_int(self.sum_order))
        u = self._arr(u); p = self._arr(p)
        out = np.empty_like(u)
        self.lib.orc_f_eval(C.byref(self.arch), self._p(p), self._p(u), C.c_int(u.shape[0]), self.real(t), self._p(out))
        return out

    def initdt(self, p, u0, t0, t1):
        self.lib.orc_set_sum_order(C.c_int(self.sum_order))
        u0 = self._arr(u0); p = self._arr(p)
        f0 = np.empty_like(u0)
        dt = self.lib.orc_initdt(C.byref(self.cfg), self._p(p), self._p(u0), C.c_int(u0.shape[0]), self.real(t0),
                                 self.real(t1), self._p(f0))
        return float(dt), f0

    def attempt(self, p, uprev, k1, t, dt, want_eigen=False):
        self.lib.orc_set_sum_order(C.c_int(self.sum_order))
        uprev = self._arr(uprev); k1 = self._arr(k1); p = self._arr(p)
        B = uprev.shape[0]
        kout = np.empty((self.S - 1, B, self.D), dtype=self.dtype)
        unew = np.empty_like(uprev)
        eest = self.real(0)
        eig = self.real(0)
        self.lib.orc_tsit5_attempt(C.byref(self.cfg), self._p(p), self._p(uprev), self._p(k1), C.c_int(B),
                                   self.real(t), self.real(dt), self._p(kout), self._p(unew), C.byref(eest),
                                   C.byref(eig) if want_eigen else None)
        return kout, unew, float(eest.value), float(eig.value)

    def forward(self, x, p, t0=0.0, t1=1.0, saveat=None):
        self.lib.orc_set_sum_order(C.c_int(self.sum_order))
        x = self._arr(x); p = self._arr(p)
        B = x.shape[0]
        ns = 0 if saveat is None else len(saveat)
        sa = self._arr(saveat) if ns else None
        u_out = np.empty((B, ns, self.D) if ns else (B, self.D), dtype=self.dtype)
        nfe = C.c_long(0)
        saveval = np.zeros(self.max_attempts + 1, dtype=self.dtype)
        nsv = C.c_int(0)
        log = np.zeros((self.max_attempts, 4), dtype=self.dtype)
        natt = C.c_int(0)
        rc = self.lib.orc_forward(self.h, self._p(x), self._p(p), C.c_int(B), self.real(t0), self.real(t1),
                                  self._p(sa) if ns else None, C.c_int(ns), self._p(u_out), C.byref(nfe),
                                  self._p(saveval), C.byref(nsv), self._p(log), C.byref(natt))
        return dict(rc=rc, u=u_out, nfe=nfe.value, saveval=saveval[:nsv.value].copy(),
                    steps=log[:natt.value].copy(), nattempts=natt.value)

    def set_replay(self, dtp=None, acc=None):
        """Following forwards take attempt n with proposed size dtp[n] and accept decision acc[n] (None: back to the controller)."""
        if dtp is None:
            self.lib.orc_set_replay(self.h, None, None, C.c_int(0))
            return
        d = self._arr(dtp); a = np.ascontiguousarray(acc, dtype=np.int32)
        assert d.shape == a.shape and d.ndim == 1
        self.lib.orc_set_replay(self.h, self._p(d), a.ctypes.data_as(C.c_void_p), C.c_int(len(d)))

    def steps_ext(self):
        """(n, 6) array of the last forward: t, dt, dtp_in, EEst, accepted, q."""
        out = np.zeros((self.max_attempts, 6), dtype=self.dtype)
        self.lib.orc_steps_ext.restype = C.c_int
        n = self.lib.orc_steps_ext(self.h, self._p(out), C.c_int(self.max_attempts))
        return out[:n].copy()

    def backward(self, ubar, svbar=None):
        ubar = self._arr(ubar)
        B = ubar.shape[0]
        xbar = np.empty((B, self.D), dtype=self.dtype)
        pbar = np.empty(self.P, dtype=self.dtype)
        tsb = np.zeros(2, dtype=self.dtype)
        sv = self._arr(svbar) if svbar is not None else None
        rc = self.lib.orc_backward(self.h, self._p(ubar), self._p(sv) if sv is not None else None, self._p(xbar),
                                   self._p(pbar), self._p(tsb))
        assert rc == 0, rc
        return xbar, pbar, tsb

    def tableau(self):
        a = np.zeros((self.S, self.S)); c = np.zeros(self.S); bt = np.zeros(self.S)
        self.lib.orc_tableau_of(C.c_int(self.solver), self._p(a), self._p(c), self._p(bt))
        return a, c, bt

    def dense_weights(self, theta):
        b = np.zeros(7)
        self.lib.orc_dense_weights_of(C.c_int(self.solver), C.c_double(theta), self._p(b))
        return b


def glorot_params(arch, rng, dtype=np.float32, scale=1.0):
    """Flux 0.11 Dense default init: W ~ U(+-sqrt(6/(in+out))), b = 0; destructure order."""
    parts = []
    for l in range(arch.n_layers):
        ine = arch.dims[l] + (1 if arch.time_dep else 0)
        o = arch.dims[l + 1]
        lim = scale * np.sqrt(6.0 / (ine + o))
        W = rng.uniform(-lim, lim, size=(ine, o))  # column-major (o x ine) == row-major (ine, o)
        parts.append(W.reshape(-1))
        parts.append(np.zeros(o))
    return np.concatenate(parts).astype(dtype)
