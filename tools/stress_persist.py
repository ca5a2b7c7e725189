"""Stress of the in-kernel hand-off: many solves of random batch sizes, persistent kernels against the 7-launch kernels, bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
rng = np.random.default_rng(0)
os.environ["RNDE_WGRAD_SIDE"] = "0"     # same GEMM partition on both paths: p-bar then checks the tape bit for bit
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
arch, p, x = _setup("mnist", 512, 5, 2.0)
nodes = {}
for persist in ("1", "0"):
    os.environ["RNDE_PERSIST"] = persist
    nodes[persist] = Node(_cfg(arch, 512, reltol=1e-6, abstol=1e-6, col_tile=16, max_attempts=256))
bad = 0
t0 = time.time()
for it in range(N):
    B = int(rng.choice([16, 48, 64, 100, 128, 256, 300, 512]))
    xs = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ps = (p * (1.0 + 0.05 * rng.standard_normal())).astype(np.float32)
    out = {}
    for k, n in nodes.items():
        g = n.forward(xs, ps, keep_tape=True)
        gx, gp, gt = n.backward(np.ones_like(xs), np.full(len(g["saveval"]), 2.0, dtype=np.float32))
        out[k] = (g["u"], g["saveval"], gx, gp, gt, g["nfe"])
    same = all(np.array_equal(a, b) for a, b in zip(out["1"][:5], out["0"][:5])) and out["1"][5] == out["0"][5]
    if not same:
        bad += 1
        print("MISMATCH at iteration", it, "B", B)
print(f"{N} solves, {bad} mismatches, {time.time() - t0:.1f} s, launches per attempt now: persist handle {nodes['1'].L.rnde_node_launches_per_attempt(nodes['1'].h)}")

# ---- second part: the weight-gradient launches on the side stream underneath the persistent sweep (RNDE_WGRAD_SIDE, read per call):
# everything but p-bar bit-identical to the run without them, p-bar equal up to the summation order, and no hand-off abandoned
bad = 0
n1 = nodes["1"]
for it in range(N):
    B = int(rng.choice([128, 256, 512, 512]))
    xs = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ps = (p * (1.0 + 0.05 * rng.standard_normal())).astype(np.float32)
    out = {}
    for side in ("0", "50"):
        os.environ["RNDE_WGRAD_SIDE"] = side
        g = n1.forward(xs, ps, keep_tape=True)
        gx, gp, gt = n1.backward(np.ones_like(xs), np.full(len(g["saveval"]), 2.0, dtype=np.float32))
        out[side] = (g["u"], g["saveval"], gx, gt, gp)
    same = all(np.array_equal(a, b) for a, b in zip(out["0"][:4], out["50"][:4]))
    close = np.abs(out["0"][4] - out["50"][4]).max() <= 1e-5 * np.abs(out["0"][4]).max()
    if not (same and close):
        bad += 1
        print("SIDE MISMATCH at iteration", it, "B", B, same, close)
print(f"side stream: {N} solves, {bad} mismatches, launches per attempt now: {n1.L.rnde_node_launches_per_attempt(n1.h)}")
