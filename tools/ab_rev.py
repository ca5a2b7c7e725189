"""A/B of the reversed attempt: [AB_B=4096] python tools/ab_rev.py LIB_A LIB_B ... (same box, alternating; fixed weights, HIP events of the library)."""
import subprocess, sys, os
code = r'''
import ctypes as C, sys, torch
sys.path.insert(0, '.')
import bench, regneuralde_jl_amd as rn
from regneuralde_jl_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
import os
B = int(os.environ.get("AB_B", "512"))
model = bench.build_model(rn, dev, B)
g = torch.Generator().manual_seed(1999)
x = torch.rand(B, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].to(dev)
rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
h = model.node._acquire(x.reshape(B, -1), True)
L.rnde_node_set_timing(h.ptr, 1)
fa = rs = rr = n = 0
for _ in range(8):
    rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    L.rnde_node_timing(h.ptr, C.byref(a), C.byref(b), C.byref(c))
    fa += a.value; rs += b.value; rr += c.value; n += int(L.rnde_node_last_attempts(h.ptr))
print("fwd %.2f rev %.2f us per attempt, rest %.3f ms" % (1e3 * fa / n, 1e3 * rs / n, rr / 8))
'''
for rep in range(3):
    for l in sys.argv[1:]:
        env = dict(os.environ, RNDE_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(os.path.basename(l), out.stdout.strip() or out.stderr[-400:], flush=True)
