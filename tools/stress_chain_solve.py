"""Stress of the chain engine's one-launch solve (rnde_chainmw.h MW_SOLVE) and one-launch reverse sweep (rnde_bchainmw.h SWEEP): many solves of random batch sizes (<= 512 columns = <= 32 workgroups
on one XCD), tolerances and weight scales (rejected steps included), against the one-launch-per-attempt paths (RNDE_CHAIN_SOLVE=0, RNDE_CHAIN_BSWEEP=0), forward
and -- through tape and slab -- reverse, bit for bit; no meeting may time out.     python tools/stress_chain_solve.py [N=300]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_chain import _setup, _cfg, _DEFAULT_TILE
from tests.util import Node
_DEFAULT_TILE[0] = 65
rng = np.random.default_rng(2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
MAXB = int(os.environ.get("STRESS_MAXB", "512"))      # STRESS_MAXB=4096: the throughput mode (more than 32 workgroups, agent-scope meeting)
arch, p, x = _setup("latent", MAXB, 5, 1.0)
bad = 0
t0 = time.time()
nodes = {}
for tol in (1e-2, 1e-4, 1.4e-8):
    for one in ("1", "0"):
        os.environ["RNDE_CHAIN_SOLVE"] = one
        os.environ["RNDE_CHAIN_BSWEEP"] = one
        nodes[(tol, one)] = Node(_cfg(arch, MAXB, reltol=tol, abstol=tol, max_attempts=512))
for it in range(N):
    tol = (1e-2, 1e-4, 1.4e-8)[it % 3]
    B = int(rng.integers(1, MAXB + 1))
    xs = rng.standard_normal((B, 20)).astype(np.float32)
    ps = (p * (1.0 + 1.0 * rng.random())).astype(np.float32)
    out = {}
    for one in ("1", "0"):
        n = nodes[(tol, one)]
        try:
            g = n.forward(xs, ps, keep_tape=True)
        except Exception as e:      # (max_attempts: both paths must say so)
            out[one] = (np.zeros(1),) * 6 + (str(e),)
            continue
        gx, gp, gt = n.backward(np.ones_like(xs), np.full(len(g["saveval"]), 2.0, dtype=np.float32))
        out[one] = (g["u"], g["saveval"], g["steps"], gx, gp, gt, g["nfe"])
    same = all(np.array_equal(a, b) for a, b in zip(out["1"][:6], out["0"][:6])) and out["1"][6] == out["0"][6]
    if not same:
        bad += 1
        print("MISMATCH at iteration", it, "B", B, "tol", tol, "nfe", out["1"][6], out["0"][6])
fb = [int(n.L.rnde_node_fallback_count(n.h)) for n in nodes.values()]
print(f"{N} solves, {bad} mismatches, {time.time() - t0:.1f} s, fallbacks {fb}")
sys.exit(1 if bad or any(fb) else 0)
