import numpy as np, sys, torch
sys.path.insert(0, '.')
import regneuralde_jl_amd as rn
from tests.test_gpu_chain import _cfg
from tests.util import Node, Oracle
from oracle.oracle import arch_latent
g = torch.Generator().manual_seed(2)
dyn = rn.LatentGenDynamics(generator=g)
p = rn.destructure(dyn).numpy()
z0 = torch.randn(24, 20, generator=g).numpy()
arch = arch_latent()
grid = np.linspace(0, 1, 49).astype(np.float32)
for sa in (None, grid):
    o64 = Oracle(arch, np.float64, reltol=1e-4, abstol=1e-4, reg_kind=1).forward(z0, p, saveat=sa)
    o32 = Oracle(arch, np.float32, reltol=1e-4, abstol=1e-4, reg_kind=1).forward(z0, p, saveat=sa)
    n = Node(_cfg(arch, 24, reltol=1e-4, abstol=1e-4))
    got = n.forward(z0, p) if sa is None else n.forward_saveat(z0, p, sa)
    print("nfe", got["nfe"], o64["nfe"], o32["nfe"])
    if sa is None: print(got["steps"]); 
    print(o64["steps"]); print(o32["steps"])
