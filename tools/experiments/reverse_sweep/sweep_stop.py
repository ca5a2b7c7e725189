import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node
B, scale, reg = 64, 3.0, 1
arch, p, x = _setup("mnist", B, 5, scale)
os.environ["RNDE_WGRAD_SIDE"] = "0"
tol = 1e-3
for k in (1, 2, 3, 4, 5, 6):
    os.environ["RNDE_DBG_REV_STOP"] = str(k)
    outs = []
    for one in ("1", "0"):
        os.environ["RNDE_STAGE_SWEEP"] = one
        node = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16, max_attempts=160, regularize=reg, wgrad_side_pct=-1))
        got = node.forward(x, p, keep_tape=True)
        ubar = np.random.default_rng(9).standard_normal(got["u"].shape).astype(np.float32)
        svbar = np.random.default_rng(10).standard_normal(len(got["saveval"])).astype(np.float32)
        gx, gp, gt = node.backward(ubar, svbar)
        outs.append(gx)
        node.close()
    a, b = outs
    d = np.abs(a - b)
    print("U after", k, "attempts: equal", np.array_equal(a, b), "maxdiff", d.max(), "ref", np.abs(b).max())
