"""Attempted-step time of the chain engine's multi-wave kernels for a few shapes, register-stationary weights (default) against the
LDS-table kernels (RNDE_CHAIN_REG=0 RNDE_CHAIN_LAT=0): python tools/ab_chain.py"""
import subprocess, sys, os
code = r'''
import ctypes as C, sys
sys.path.insert(0, '.')
from tests.test_gpu_chain import _setup
from tests.test_gpu_forward import _cfg
from tests.util import Node
for kind in ("latent", "chain3", "wide", "small"):
    arch, p, x = _setup(kind, 512, 7, 1.0)
    n = Node(_cfg(arch, 512, max_attempts=64, col_tile=65))
    us = C.c_float(0)
    n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), 512, 100, C.byref(us), None)
    print(kind, "%.1f" % us.value, end="  ")
print()
'''
for tag, env in (("registers", {}), ("lds-table", {"RNDE_CHAIN_REG": "0", "RNDE_CHAIN_LAT": "0"})):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    print(tag, out.stdout.strip() or out.stderr[-400:], flush=True)
