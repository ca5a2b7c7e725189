"""Cycle stamps of attempt RNDE_DIAG_FWD of a REAL solve (not the forced attempt of the micro-benchmark):
RNDE_DIAG_FWD=12 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so python tools/diag_solve.py   (build: tools/build_diag.sh)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = 512
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, reltol=1.4e-8, abstol=1.4e-8, max_attempts=128, col_tile=16))
for rep in range(3):
    r = n.forward(x, p, 0.0, 1.0, keep_tape=True)
print("attempts", r["nattempts"])
