"""Where does the regulariser's NOISE gradient come from?  ubar = 0, cotangent lambda / n on every saved EEst*dt (lambda = 100), B = 512, tol 1.4e-8:
device in both matrix modes, and the CPU oracles (fp32 sequential, fp32 in the fp32-MFMA order, fp64) replaying each device run's step sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import Node, Oracle, arch_mnist, glorot_params, make_cfg
B, tol = 512, 1.4e-8
rng = np.random.default_rng(31)
arch = arch_mnist()
p = glorot_params(arch, rng, np.float32, 1.0)
x = rng.uniform(0, 1, (B, 784)).astype(np.float32)
ubar0 = np.zeros((B, 784), np.float32)
ubar = (rng.standard_normal((B, 784)) / B).astype(np.float32)
for mode in (0, 1):
    node = Node(make_cfg([784, 100, 784], ["tanh", "tanh"], B, reltol=tol, abstol=tol, max_attempts=200, regularize=1), matrix_mode=mode)
    got = node.forward(x, p, keep_tape=True)
    st = got["steps"]; n = len(got["saveval"])
    svbar = np.full(n, 100.0 / n, np.float32)
    _, g_dev, _ = node.backward(ubar0, svbar)
    got = node.forward(x, p, keep_tape=True)
    _, g_sig, _ = node.backward(ubar, None)
    print(f"mode {mode}: attempts {len(st)}  mean EEst {st[:, 2].mean():.3f}  mean dt {st[:, 1].mean():.4f}  |g_reg(device)| {np.linalg.norm(g_dev):.3e} max {np.abs(g_dev).max():.3e}   |g_signal| {np.linalg.norm(g_sig):.3e} max {np.abs(g_sig).max():.3e}")
    for name, dt_, so in (("f32 seq", np.float32, 0), ("f32 devorder", np.float32, 3), ("f64", np.float64, 0)):
        o = Oracle(arch, dt_, tol, tol, reg_kind=1, max_attempts=200, sum_order=so)
        o.set_replay(st[:, 1].astype(dt_), st[:, 3].astype(np.int32))
        r = o.forward(x.astype(dt_), p.astype(dt_))
        _, g, _ = o.backward(ubar0.astype(dt_), svbar.astype(dt_))
        print(f"     oracle {name:12s} replaying these steps: mean EEst {r['steps'][:, 2].mean():.3e}  |g_reg| {np.linalg.norm(g):.3e} max {np.abs(g).max():.3e}   cos(device, oracle) {float(g_dev.astype(np.float64) @ g.astype(np.float64) / (np.linalg.norm(g_dev) * np.linalg.norm(g) + 1e-300)):.3f}")
    node.close()
