"""Round-5 stress: (1) the one-launch forward solve with the controller on one wave against the launch-per-attempt path, bit for bit, over random
batches and weights; (2) large-batch reverse sweeps with the partial sums formed once (rnde_bpart_reduce_kernel) against the per-workgroup form, bit
for bit; (3) the SDE stiffness regulariser on SOSRI2: repeated solves + reverse passes on fresh noise, finite results, the saved values equal to the
norms the step log carries.   python tools/stress_r05.py [N=150]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node, NsdeNode
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150

# (1)
arch, p, x = _setup("mnist", 512, 5, 2.0)
nodes = {}
for solve in ("1", "0"):
    os.environ["RNDE_STAGE_SOLVE"] = solve
    nodes[solve] = Node(_cfg(arch, 512, reltol=1e-6, abstol=1e-6, col_tile=16, max_attempts=256))
os.environ.pop("RNDE_STAGE_SOLVE")
bad, t0 = 0, time.time()
for it in range(N):
    B = int(rng.choice([16, 48, 64, 100, 128, 256, 300, 512]))
    xs = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ps = (p * (1.0 + 0.05 * rng.standard_normal())).astype(np.float32)
    out = {}
    for k, n in nodes.items():
        g = n.forward(xs, ps, keep_tape=True)
        gx, gp, gt = n.backward(np.ones_like(xs), np.full(len(g["saveval"]), 2.0, dtype=np.float32))
        out[k] = (g["u"], g["saveval"], gx, gp, gt, g["nfe"])
    if not (all(np.array_equal(a, b) for a, b in zip(out["1"][:5], out["0"][:5])) and out["1"][5] == out["0"][5]):
        bad += 1
        print("MISMATCH (one-launch solve) at iteration", it, "B", B)
print(f"(1) {N} solves, {bad} mismatches, {time.time() - t0:.1f} s; one-launch solves: {nodes['1'].L.rnde_node_one_launch_solves(nodes['1'].h)}")
for n in nodes.values():
    n.close()

# (2)
B = 2048
arch, p, x = _setup("mnist", B, 8, 2.0)
bad, t0 = 0, time.time()
node = Node(_cfg(arch, B, reltol=1e-5, abstol=1e-5, col_tile=16, max_attempts=64))
for it in range(max(4, N // 15)):
    xs = rng.uniform(0, 1, (B, 784)).astype(np.float32)
    ps = (p * (1.0 + 0.05 * rng.standard_normal())).astype(np.float32)
    ub = rng.standard_normal((B, 784)).astype(np.float32) / B
    res = []
    for off in (None, "1"):
        if off:
            os.environ["RNDE_NO_BPART_REDUCE"] = off
        else:
            os.environ.pop("RNDE_NO_BPART_REDUCE", None)
        f = node.forward(xs, ps, keep_tape=True)
        res.append(node.backward(ub, np.linspace(0.5, 1.5, len(f["saveval"])).astype(np.float32)))
    if not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(*res)):
        bad += 1
        print("MISMATCH (partial sums formed once) at iteration", it)
os.environ.pop("RNDE_NO_BPART_REDUCE", None)
print(f"(2) {max(4, N // 15)} sweeps at B = {B}, {bad} mismatches, {time.time() - t0:.1f} s")
node.close()

# (3)
from tests.test_gpu_nsde import _setup as s_setup, _cfg as s_cfg
bad, t0 = 0, time.time()
for mw in ("1", "0"):
    os.environ["RNDE_SDE_MW"] = mw
    drift, diff, p, x, noise = s_setup("nsde", 512, 3, 257, scale=1.5, dscale=0.8)
    node = NsdeNode(s_cfg(drift, diff, 512, solver="SOSRI2", regularize=2, max_attempts=256))
    for it in range(N):
        nz = rng.standard_normal(noise.shape).astype(np.float32)
        got = node.forward(x, p, nz, keep_tape=True)
        sv = got["saveval"]
        xb, pb = node.backward(np.ones_like(x) / 512, np.full(len(sv), 0.1 / len(sv), np.float32))
        ok = np.isfinite(got["u"]).all() and np.isfinite(xb).all() and np.isfinite(pb).all() and (sv[1:] > 0).all() and len(sv) == int(got["steps"][:, 3].sum()) + 1
        if not ok:
            bad += 1
            print("BAD SDE stiff solve at iteration", it, "mw", mw)
    node.close()
print(f"(3) 2 x {N} SDE solves with the stiffness regulariser, {bad} bad, {time.time() - t0:.1f} s")
