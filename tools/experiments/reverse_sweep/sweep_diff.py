"""debug: one-launch reverse sweep vs launch-per-attempt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node

B, tol, scale, reg = int(os.environ.get("DB", "64")), 1e-3, 3.0, 1
arch, p, x = _setup("mnist", B, 5, scale)
os.environ["RNDE_WGRAD_SIDE"] = "0"
outs = []
for one in ("1", "0"):
    os.environ["RNDE_STAGE_SWEEP"] = one
    node = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16, max_attempts=160, regularize=reg, wgrad_side_pct=-1))
    got = node.forward(x, p, keep_tape=True)
    ubar = np.random.default_rng(9).standard_normal(got["u"].shape).astype(np.float32)
    svbar = np.random.default_rng(10).standard_normal(len(got["saveval"])).astype(np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    outs.append((got, gx, gp, gt))
    node.close()
a, b = outs
print("attempts", a[0]["nattempts"], "seg", os.environ.get("RNDE_SWEEP_SEG"))
for i, nm in ((1, "xbar"), (2, "pbar"), (3, "tbar")):
    d = np.abs(a[i] - b[i])
    print(nm, "equal", np.array_equal(a[i], b[i]), "max diff", d.max(), "ref max", np.abs(b[i]).max())
    if nm == "xbar" and d.max() > 0:
        bad = np.argwhere(d > 0)
        print(" bad cols", np.unique(bad[:, 0])[:20], "bad rows", np.unique(bad[:, 1])[:20], "count", len(bad), "of", d.size)
