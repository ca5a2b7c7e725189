import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import Node, Oracle
from tests.test_gpu_forward import _setup, _cfg
np.set_printoptions(linewidth=200, precision=6)
for kind, B, tol in [("test_node", 1, 1e-4), ("mnist", 32, 1e-5)]:
    arch, p, x = _setup(kind, B, 3)
    ref = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1).forward(x, p)
    ref64 = Oracle(arch, np.float64, reltol=tol, abstol=tol, reg_kind=1).forward(x, p)
    got = Node(_cfg(arch, B, reltol=tol, abstol=tol)).forward(x, p)
    print(kind, "device steps (t, dt, EEst, acc)\n", got["steps"])
    print("oracle f32\n", ref["steps"])
    print("oracle f64\n", ref64["steps"])
