"""A/B of the forward attempt kernel: python tools/ab_attempt.py LIB_A LIB_B ... (same box, alternating, child process per run)."""
import subprocess, sys, os
code = r'''
import ctypes as C, sys
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
import os
B = int(os.environ.get('AB_B', '512'))
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=16 if B > 1024 else 64, col_tile=16))
us = C.c_float(0); ust = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 300, C.byref(us), None)
n.L.rnde_bench_attempt_taped(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 300, C.byref(ust), None)
print("%.2f %.2f" % (us.value, ust.value))
'''
libs = sys.argv[1:]
for rep in range(3):
    for l in libs:
        env = dict(os.environ, RNDE_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(l), out.stdout.strip() or out.stderr[-300:], flush=True)
