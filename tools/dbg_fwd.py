import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import Node, Oracle
from tests.test_gpu_forward import _setup, _cfg
np.set_printoptions(linewidth=200, precision=6, suppress=True)
for kind, B, tol, scale, t1, seed in [("test_node", 3, 1e-2, 10.0, 3.0, 0)]:
    arch, p, x = _setup(kind, B, seed, scale)
    o = Oracle(arch, np.float32, reltol=tol, abstol=tol, reg_kind=1)
    ref = o.forward(x, p, 0.0, t1)
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol))
    got = node.forward(x, p, 0.0, t1)
    n = min(len(got["steps"]), len(ref["steps"]))
    print(np.hstack([got["steps"][:n], ref["steps"][:n]]))
    # single attempt comparison at the first diverging step
    k1 = o.f_eval(p, x, 0.0)
    for (t, dt) in [(0.0, float(ref["steps"][0,1])), (0.0, 0.2)]:
        kr, ur, er, _ = o.attempt(p, x, k1, t, dt)
        kd, ud, ed = node.attempt(x, k1, p, t, dt)
        print("attempt dt", dt, "max|dk|", np.abs(kd-kr).max(0).max(1) if False else np.abs(kd-kr).reshape(6,-1).max(1), "eest", ed, er)
    fd = node.feval(x, p, 0.3); fr = o.f_eval(p, x, 0.3)
    print("feval diff", np.abs(fd-fr).max(), np.abs(fr).max())
