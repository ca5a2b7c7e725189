import os, sys, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
arch, p, x = _setup("mnist", B, 5, 1.0)
from tests.util import Oracle
k1 = Oracle(arch, np.float32).f_eval(p, x, 0.1)
res = []
for persist in ("1", "0"):
    os.environ["RNDE_PERSIST"] = persist
    n = Node(_cfg(arch, B, col_tile=16))
    kout, unew, eest = n.attempt(x, k1, p, 0.1, 0.03)
    res.append((kout, unew, eest))
a, b = res
print("eest", a[2], b[2])
kd = np.abs(a[0] - b[0])
print("kout shape", a[0].shape)
for s in range(6):
    blk = kd.reshape(6, -1)[s] if kd.ndim == 1 else kd[s]
    print("k%d max diff" % (s + 2), blk.max(), "nonzero", int((blk > 0).sum()), "of", blk.size)
print("unew diff", np.abs(a[1] - b[1]).max())
d3 = (a[0][1] != b[0][1])   # k3: (B, D)
print("k3: differing entries per column (first 16 cols):", d3.sum(axis=1)[:16], " per row tile of 16 (first 10):", d3.reshape(d3.shape[0], -1, 16).sum(axis=(0, 2))[:10])
