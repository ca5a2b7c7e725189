"""A fixed reverse pass for timing the weight-gradient kernels under rocprofv3: one natural forward solve at the headline shape (B = 512, tol 1.4e-8, Glorot
weights x 3), then `--reps` identical reverse passes -- the same evaluations in every library variant, whatever its gradients are worth (tools/gpu_wgrad_times.sh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node
arch, p, x = _setup("mnist", args.batch, 7, 3.0)
n = Node(_cfg(arch, args.batch, reltol=1.4e-8, abstol=1.4e-8, regularize=1, max_attempts=200, col_tile=16, persist=1, wgrad_side_pct=30))
for _ in range(args.reps):
    g = n.forward(x, p, keep_tape=True)
    out = n.backward(np.ones_like(x), np.full(len(g["saveval"]), 1.0, dtype=np.float32))
print("attempts", g["nattempts"], "p-bar max", float(np.abs(out[1]).max()))
