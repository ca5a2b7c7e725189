"""First contact of the X3 one-launch solve (csrc/rnde_x3.h) with the MI355X: the same natural run with the fp32-MFMA kernel and with the matrix-core kernel,
each against the fp64 restatement replayed along ITS OWN step sequence; attempts, error of u_end, time per solve.   python tools/x3_check.py [B] [tol]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.util import Node, Oracle, arch_mnist, glorot_params, make_cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1.4e-8
rng = np.random.default_rng(11)
arch = arch_mnist()
for scale in (1.0, 2.5):
    p = glorot_params(arch, rng, np.float32, scale)
    x = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    res = {}
    for x3 in (0, 1):
        os.environ["RNDE_X3"] = str(x3)
        node = Node(make_cfg([784, 100, 784], ["tanh", "tanh"], B, reltol=tol, abstol=tol, regularize=1, max_attempts=400))
        got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
        st = got["steps"]
        o64 = Oracle(arch, np.float64, tol, tol, reg_kind=1, max_attempts=400)
        o64.set_replay(st[:, 1].astype(np.float64), st[:, 3].astype(np.int32))
        r64 = o64.forward(x.astype(np.float64), p.astype(np.float64))
        err = float(np.abs(got["u"] - r64["u"]).max() / np.abs(r64["u"]).max())
        ubar = (rng.standard_normal((B, 784)) / B).astype(np.float32)
        xb, pb, _ = node.backward(ubar, None)
        g64 = o64.backward(ubar.astype(np.float64), None)
        eg = float(np.abs(pb - g64[1]).max() / np.abs(g64[1]).max())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            node.forward(x, p, 0.0, 1.0, keep_tape=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        one = node.L.rnde_node_one_launch_solves(node.h)
        res[x3] = (len(st), int(st[:, 3].sum()), err, eg, ms)
        print(f"scale {scale} x3={x3}: attempts {len(st)} accepted {int(st[:, 3].sum())} nfe {got['nfe']}  |u - fp64(replay)| {err:.2e}  p-bar vs fp64 {eg:.2e}  "
              f"{ms:.3f} ms per taped solve = {1e3 * ms / len(st):.2f} us per attempt (one-launch solves {one})", flush=True)
        node.close()
    print(f"   speed-up per attempt {res[0][4] / res[0][0] / (res[1][4] / res[1][0]):.2f}x, per solve {res[0][4] / res[1][4]:.2f}x")
# natural-run statistics against the oracles (fp32 sequential, fp32 device order, fp64)
p = glorot_params(arch, np.random.default_rng(5), np.float32, 1.0)
x = np.random.default_rng(6).uniform(0, 1, (64, 784)).astype(np.float32)
for name, o in (("oracle f32 sequential", Oracle(arch, np.float32, tol, tol, reg_kind=1, max_attempts=400)), ("oracle f32 device order", Oracle(arch, np.float32, tol, tol, reg_kind=1, max_attempts=400, sum_order=3)),
                ("oracle f64", Oracle(arch, np.float64, tol, tol, reg_kind=1, max_attempts=400))):
    r = o.forward(x.astype(o.dtype), p.astype(o.dtype))
    print(f"B = 64 natural run, {name}: attempts {r['nattempts']}")
for x3 in (0, 1):
    os.environ["RNDE_X3"] = str(x3)
    node = Node(make_cfg([784, 100, 784], ["tanh", "tanh"], 64, reltol=tol, abstol=tol, regularize=1, max_attempts=400))
    print(f"B = 64 natural run, device x3={x3}: attempts {node.forward(x, p)['nattempts']}")
    node.close()
