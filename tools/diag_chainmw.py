"""Cycle stamps of one multi-wave chain-engine attempt (config 4 shape): RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so python tools/diag_chainmw.py"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_chain import _setup
from tests.test_gpu_forward import _cfg
from tests.util import Node
B = 512
arch, p, x = _setup("latent", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=64, col_tile=65))
us = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 50, C.byref(us), None)
print("attempt us", us.value)
