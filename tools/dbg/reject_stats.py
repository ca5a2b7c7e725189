"""How many attempted steps of the bench's solve are rejected, at the reference tolerance (steps[:, 3] = accepted flag; EEst in [:, 2])."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import regneuralde_jl_amd as rn
from bench import build_model
dev = torch.device("cuda", 0)
model = build_model(rn, dev, 512)
g = torch.Generator().manual_seed(1999)
x = torch.rand(512, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (512,), generator=g)].to(dev)
opt = rn.FluxOptimiser(model.trainable())
from regneuralde_jl_amd import _lib
import ctypes as C
L = _lib.lib()
for it in range(30):
    rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
    opt.step()
    if it % 5 == 4:
        h = model.node._acquire(x.reshape(512, -1), True)
        steps = (C.c_float * (4 * 160))(); natt = C.c_int32(0)
        L.rnde_node_steps(h.ptr, steps, 160, C.byref(natt))
        s = np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4)
        acc = s[:, 3] != 0
        print(f"step {it}: attempts {natt.value}, accepted {int(acc.sum())}, rejected {int((~acc).sum())}, EEst of the rejected: {np.round(s[~acc, 2], 2)}")
