"""debug: one-launch solve vs launch-per-attempt, field by field"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node
import ctypes as C
from regneuralde_jl_amd import _lib

for (B, tol, scale, reg) in [(200, 1e-3, 3.0, 3), (37, 1e-4, 2.0, 2), (200, 1e-3, 3.0, 1)]:
    arch, p, x = _setup("mnist", B, 5, scale)
    outs = []
    for one in ("1", "0"):
        os.environ["RNDE_STAGE_SOLVE"] = one
        node = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16, max_attempts=160, regularize=reg))
        got = node.forward(x, p, keep_tape=True)
        meta = (C.c_float * (16 * 200))()
        outs.append(got)
        node.close()
    a, b = outs
    print("case", B, tol, reg, "nfe", a["nfe"], b["nfe"], "u equal", np.array_equal(a["u"], b["u"]))
    print(" steps equal", np.array_equal(a["steps"], b["steps"]))
    if not np.array_equal(a["steps"], b["steps"]):
        d = np.argwhere(a["steps"] != b["steps"])
        print(d[:10], a["steps"][d[0][0]], b["steps"][d[0][0]])
    print(" saveval", a["saveval"].view(np.uint32) - b["saveval"].view(np.uint32))
