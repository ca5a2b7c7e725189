"""Per-attempt time of the chain engine INSIDE real solves (latent shape, B = 512, tol 1.4e-8, 49 save points, taped), forward sweep / reverse sweep
by HIP events (rnde_node_set_timing): python tools/ab_chain_solve.py LIB_A LIB_B ...   (same box, alternating, a child process per run)"""
import subprocess, sys, os
code = r'''
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
fa, rs = [], []
for rep in range(8):
    r = n.forward_saveat(x, p, sa, 0.0, 1.0, keep_tape=True)
    ub = np.ones_like(r["u"])
    n.backward(ub, np.full(len(r["saveval"]), 1.0, dtype=np.float32))
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    n.L.rnde_node_timing(n.h, C.byref(a), C.byref(b), C.byref(c))
    att = n.L.rnde_node_last_attempts(n.h)
    if rep >= 2: fa.append(1e3 * a.value / att); rs.append(1e3 * b.value / att)
print("attempts %d  fwd %.2f us/attempt  rev %.2f us/attempt" % (att, sum(fa) / len(fa), sum(rs) / len(rs)))
'''
libs = sys.argv[1:]
for rep in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RNDE_LIB=os.path.abspath(l)), capture_output=True, text=True)
        print(os.path.basename(l), out.stdout.strip() or out.stderr[-500:], flush=True)
