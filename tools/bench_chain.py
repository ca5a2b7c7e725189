"""Config 4: latent-ODE dynamics on the chain engine -- attempt time, forward / reverse time."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_chain import _setup, _cfg
from tests.util import Node
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
arch, p, x = _setup("latent", B, 7, 1.0)
sa = np.linspace(0, 1, 49).astype(np.float32)
n = Node(_cfg(arch, B, max_attempts=512, col_tile=int(sys.argv[2]) if len(sys.argv) > 2 else 64))
xd, pd = n.dev(x), n.dev(p)
us = C.c_float(0)
n.L.rnde_bench_attempt(n.h, xd.data_ptr(), pd.data_ptr(), B, 200, C.byref(us), None)
print(f"attempt: {us.value:.2f} us")
T = len(sa)
u = torch.empty((B, T, 20), device="cuda"); nfe = C.c_int64(0); nsv = C.c_int32(0); sv = (C.c_float * 513)(); saa = (C.c_float * T)(*sa.tolist())
ub = torch.randn(B, T, 20, device="cuda"); xb = torch.empty(B, 20, device="cuda"); pb = torch.empty(p.size, device="cuda"); tsb = (C.c_float * 2)()
def fwd(tape):
    st = n.L.rnde_node_forward_saveat(n.h, xd.data_ptr(), pd.data_ptr(), B, 0.0, 1.0, saa, T, u.data_ptr(), C.byref(nfe), sv, C.byref(nsv), tape, None)
    assert st == 0, st
def bwd():
    st = n.L.rnde_node_backward(n.h, ub.data_ptr(), None, xb.data_ptr(), pb.data_ptr(), tsb, None)
    assert st == 0, st
for _ in range(3): fwd(1); bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): fwd(1)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(20): fwd(1); bwd()
torch.cuda.synchronize(); t2 = time.perf_counter()
f = (t1 - t0) / 20 * 1e3; fb = (t2 - t1) / 20 * 1e3
print(f"B={B} nfe={nfe.value} attempts={(nfe.value-3)//6}  forward {f:.3f} ms  forward+reverse {fb:.3f} ms  ({B/(fb*1e-3):.0f} samples/s)")
