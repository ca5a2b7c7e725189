import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_chain import _setup
from tests.test_gpu_forward import _cfg
from tests.util import Node
arch, p, x = _setup("latent", 512, 7, 1.0)
n = Node(_cfg(arch, 512, reltol=1.4e-8, abstol=1.4e-8, max_attempts=256))
sa = np.linspace(0, 1, 49).astype(np.float32)
n.L.rnde_node_set_timing(n.h, 1)
def t(fn, reps=8):
    fa = []
    for rep in range(reps):
        fn()
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        n.L.rnde_node_timing(n.h, C.byref(a), C.byref(b), C.byref(c))
        att = n.L.rnde_node_last_attempts(n.h)
        if rep >= 2: fa.append(1e3 * a.value / att)
    return sum(fa) / len(fa), att
print("saveat taped   %.2f us/attempt (%d attempts)" % t(lambda: n.forward_saveat(x, p, sa, 0.0, 1.0, keep_tape=True)))
print("saveat untaped %.2f us/attempt (%d attempts)" % t(lambda: n.forward_saveat(x, p, sa, 0.0, 1.0, keep_tape=False)))
print("end-state taped   %.2f us/attempt (%d attempts)" % t(lambda: n.forward(x, p, 0.0, 1.0, keep_tape=True)))
print("end-state untaped %.2f us/attempt (%d attempts)" % t(lambda: n.forward(x, p, 0.0, 1.0, keep_tape=False)))
us = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), 512, 100, C.byref(us), None)
print("forced back-to-back untaped %.2f" % us.value)
