import ctypes as C, sys, os
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = 4096
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=16, col_tile=16))
us = C.c_float(0); ust = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 100, C.byref(us), None)
n.L.rnde_bench_attempt_taped(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 100, C.byref(ust), None)
print("%.1f %.1f" % (us.value, ust.value))
