"""Cycle stamps of one stage-engine attempt: RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so python tools/diag_stage.py (build: tools/build_diag.sh)"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = 512
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=64, col_tile=16))
us = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 100, C.byref(us), None)
print("attempt us", us.value)
