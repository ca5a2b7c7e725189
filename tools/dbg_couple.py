import ctypes as C, threading, sys
sys.path.insert(0,'.')
import numpy as np
from regneuralde_jl_amd import _lib
from tests.test_gpu_forward import _cfg, _setup
from tests.util import Node
L=_lib.lib()
B=64; tol=1e-3
arch,p,x=_setup("small",B,21,5.0)
ref=Node(_cfg(arch,B,reltol=tol,abstol=tol,col_tile=16)).forward(x,p)
comms=(C.c_void_p*2)()
assert L.rnde_comm_create_local_group(2,0,comms)==0
nodes=[Node(_cfg(arch,32,reltol=tol,abstol=tol,col_tile=16)).own_stream() for _ in range(2)]
for n,c in zip(nodes,comms): _lib.check(n.h,L.rnde_node_set_coupling(n.h,C.c_void_p(c),B))
out=[None,None]
def work(r): out[r]=nodes[r].forward(x[32*r:32*r+32],p)
th=[threading.Thread(target=work,args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
np.set_printoptions(precision=6, linewidth=200)
print("ref steps (t, dt, eest, acc):\n", ref["steps"][:8])
print("rank0:\n", out[0]["steps"][:8])
print("rank1:\n", out[1]["steps"][:8])
# uncoupled shards for comparison
for r in range(2):
    u=Node(_cfg(arch,32,reltol=tol,abstol=tol,col_tile=16)).forward(x[32*r:32*r+32],p)
    print("uncoupled shard",r, u["steps"][:3])
