import os, sys, time, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
arch, p, x = _setup("mnist", B, 5, 3.0)
outs = []
for persist in ("1", "0", "1"):
    os.environ["RNDE_PERSIST"] = persist
    n = Node(_cfg(arch, B, reltol=1e-3, abstol=1e-3, col_tile=16))
    got = n.forward(x, p)
    got2 = n.forward(x, p)
    print(persist, "nfe", got["nfe"], got2["nfe"], "u[0,:3]", got["u"][0, :3], "same on repeat:", np.array_equal(got["u"], got2["u"]))
    outs.append(got["u"])
print("persist vs multi max diff", np.abs(outs[0] - outs[1]).max(), " persist vs persist(new handle)", np.abs(outs[0] - outs[2]).max())
