import numpy as np, sys
sys.path.insert(0, '.')
from tests.test_gpu_chain import _setup, _cfg
from tests.util import Node, Oracle
arch, p, x = _setup("latent", 4, 3, 2.0)
ref = Oracle(arch, np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1).forward(x, p)
got = Node(_cfg(arch, 4, reltol=1e-3, abstol=1e-3)).forward(x, p)
np.set_printoptions(linewidth=250, precision=2)
print(np.abs(got["u"] - ref["u"]))
print(got["steps"]); print(ref["steps"])
print(got["saveval"], ref["saveval"])
