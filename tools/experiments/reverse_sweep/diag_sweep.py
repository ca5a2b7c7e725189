"""Cycle stamps of one reversed attempt inside the one-launch reverse sweep (tools/diag_sweep.sh builds the library)."""
import sys, os
sys.path.insert(0, '.')
os.environ["RNDE_DIAG_SWEEP"] = "1"
os.environ.setdefault("RNDE_WGRAD_SIDE", "0")
import numpy as np
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
arch, p, x = _setup("mnist", 512, 7, 1.0)
n = Node(_cfg(arch, 512, max_attempts=256, col_tile=16))
for _ in range(3):
    g = n.forward(x, p, keep_tape=True)
    n.backward(np.ones_like(x), np.full(len(g["saveval"]), 1.0, dtype=np.float32))
