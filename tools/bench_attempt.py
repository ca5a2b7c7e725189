"""Time the forward step kernel alone through the C ABI (rnde_bench_attempt) for a given librnde build."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 512; ct = int(sys.argv[3]) if len(sys.argv) > 3 else 0
from regneuralde_jl_amd import _lib, build
build.LIB = lib; _lib._lib = None
from tests.util import make_cfg, Node
cfg = make_cfg([784, 100, 784], ["tanh", "tanh"], B, col_tile=ct)
node = Node(cfg)
rng = np.random.default_rng(0)
x = torch.rand(B, 784, device="cuda"); p = (torch.rand(158568, device="cuda") - 0.5) * 0.16
us = C.c_float(0)
for _ in range(2):
    _lib.check(node.h, node.L.rnde_bench_attempt(node.h, x.data_ptr(), p.data_ptr(), B, 100, C.byref(us), None))
print(f"{os.path.basename(lib)} B={B} col_tile={ct}: {us.value:.1f} us/attempt = {us.value/6:.2f} us/stage")
