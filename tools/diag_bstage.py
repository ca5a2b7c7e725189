"""Cycle stamps of one reversed stage-engine attempt: RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so python tools/diag_bstage.py"""
import sys, os
sys.path.insert(0, '.')
os.environ["RNDE_DIAG_BWD"] = "1"
import numpy as np
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
arch, p, x = _setup("mnist", 512, 7, 1.0)
n = Node(_cfg(arch, 512, max_attempts=64, col_tile=16))
g = n.forward(x, p, keep_tape=True)
n.backward(np.ones_like(x), np.full(len(g["saveval"]), 1.0, dtype=np.float32))
