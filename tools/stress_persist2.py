"""Stress of the two-column-tile forward attempt kernel (rnde_stage_persist2.h): many solves of random batch sizes (any even number of
16-column tiles up to 2048 columns, ragged last tile included), two tiles per workgroup (alternating form, RNDE_PERSIST2=1) against one
tile per workgroup (=0), forward and -- through the tape -- reverse, bit for bit; no hand-off may be abandoned.
    python tools/stress_persist2.py [N=150]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
rng = np.random.default_rng(1)
os.environ["RNDE_WGRAD_SIDE"] = "0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
arch, p, x = _setup("mnist", 2048, 5, 2.0)
nodes = {}
for two in ("1", "0"):
    os.environ["RNDE_PERSIST2"] = two
    nodes[two] = Node(_cfg(arch, 2048, reltol=1e-5, abstol=1e-5, col_tile=16, max_attempts=96))
bad = 0
t0 = time.time()
for it in range(N):
    tiles = 2 * int(rng.integers(1, 65))
    B = 16 * tiles - int(rng.integers(0, 16))           # ragged last tile
    xs = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ps = (p * (1.0 + 0.05 * rng.standard_normal())).astype(np.float32)
    out = {}
    for k, n in nodes.items():
        g = n.forward(xs, ps, keep_tape=True)
        gx, gp, gt = n.backward(np.ones_like(xs), np.full(len(g["saveval"]), 2.0, dtype=np.float32))
        out[k] = (g["u"], g["saveval"], g["steps"], gx, gp, gt, g["nfe"])
    same = all(np.array_equal(a, b) for a, b in zip(out["1"][:6], out["0"][:6])) and out["1"][6] == out["0"][6]
    if not same:
        bad += 1
        print("MISMATCH at iteration", it, "B", B)
fb = [int(n.L.rnde_node_fallback_count(n.h)) for n in nodes.values()]
print(f"{N} solves (16..2048 columns), {bad} mismatches, {time.time() - t0:.1f} s, fallbacks {fb}, launches per attempt {[int(n.L.rnde_node_launches_per_attempt(n.h)) for n in nodes.values()]}")
sys.exit(1 if bad or any(fb) else 0)
