"""Where a latent-ODE end-to-end step spends its wall time: every C-ABI call of rn.fused_latent_loss_and_grad bracketed by a device
synchronisation and a host clock (the bench's own step is run beside it, unbracketed).   python tools/dbg/latent_e2e_breakdown.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import regneuralde_jl_amd as rn
from regneuralde_jl_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
B, T = 512, 49
g = torch.Generator().manual_seed(1999)
grid = torch.linspace(0, 1, T)
model = rn.build_latent_ode(saveat=grid, regularize=True, generator=g, device=dev, max_batch=B, max_attempts=256)
data = torch.randn(B, T, 37, generator=g).to(dev)
mask = (torch.rand(B, T, 37, generator=g) < 0.3).float().to(dev)
mask[:, 0, 0] = 1.0
t_row = torch.full((B, T, 1), 1.0 / (T - 1)).to(dev); t_row[:, -1] = 0.0
opt = rn.FluxAdaMax(model.trainable())
def step():
    out = rn.fused_latent_loss_and_grad(model, data, mask, t_row, lam_r=1.0e3, lam_k=1.0, generator=None)
    opt.step()
    return out
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): o = step()
    torch.cuda.synchronize()
    print("plain step: %.3f ms  nfe %d" % ((time.perf_counter() - t0) / 20 * 1e3, o[4]))
# the same step with a clock between the calls
import regneuralde_jl_amd.timeseries as ts
names, acc = [], {}
def wrap(name):
    f = getattr(L, name)
    def w(*a):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        acc.setdefault(name, []).append((t1 - t0, t2 - t0))
        return r
    return w
class LW:
    def __getattr__(self, k):
        if k in ("rnde_latent_encode", "rnde_node_forward_saveat", "rnde_latent_decode_loss", "rnde_node_backward_async", "rnde_latent_encode_backward", "rnde_adamax_step"):
            return wrap(k)
        return getattr(L, k)
_lib_lib = _lib.lib
_lib.lib = lambda: LW()
for rep in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); acc.setdefault("whole step (bracketed)", []).append((0.0, time.perf_counter() - t0))
for k, v in acc.items():
    print("%-32s host %.3f ms   host+device %.3f ms   (n %d per step)" % (k, sum(a for a, _ in v) / 10 * 1e3, sum(b for _, b in v) / 10 * 1e3, len(v) // 10))
