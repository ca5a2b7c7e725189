"""forward + reverse time per training-shaped call at small batches, one-launch reverse sweep on / off"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node
for B in (16, 64, 128, 256, 512):
    arch, p, x = _setup("mnist", B, 5, 1.0)
    res = {}
    for sweep in ("1", "0"):
        os.environ["RNDE_STAGE_SWEEP"] = sweep
        node = Node(_cfg(arch, B, col_tile=16, max_attempts=160, regularize=1))
        got = node.forward(x, p, keep_tape=True)
        ubar = np.ones_like(got["u"]); svbar = np.ones(len(got["saveval"]), dtype=np.float32)
        node.backward(ubar, svbar)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            got = node.forward(x, p, keep_tape=True)
            node.backward(ubar, svbar)
        torch.cuda.synchronize(); res[sweep] = (time.perf_counter() - t0) / 10 * 1e3
        node.close()
    print(f"B {B}: attempts {got['nattempts']}  fwd+rev ms  sweep {res['1']:.3f}  per-attempt {res['0']:.3f}")
