"""rnde_tapes_*: several taped forwards alive at once behind one handle (tape ids) -- the reference's loop runs an NFE probe
on a fixed batch between a forward and its reverse (experiments/mnist_node.jl:245) and a plain rnde_node would drop the first tape."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fwd(L, t, x, p, keep):
    import torch
    xd, pd = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    u = torch.empty_like(xd)
    nfe, nsv, tid = C.c_int64(0), C.c_int32(0), C.c_int32(-7)
    sv = (C.c_float * 129)()
    st = L.rnde_tapes_forward(t, xd.data_ptr(), pd.data_ptr(), x.shape[0], 0.0, 1.0, u.data_ptr(), C.byref(nfe), sv, C.byref(nsv), keep, None, C.byref(tid))
    return st, u.cpu().numpy(), nfe.value, np.array(sv[:nsv.value], dtype=np.float32), tid.value


def _bwd(L, t, tid, ubar, svb, P):
    import torch
    ub = torch.from_numpy(ubar).cuda()
    xb, pb = torch.empty_like(ub), torch.empty(P, dtype=torch.float32, device="cuda")
    tsb = (C.c_float * 2)()
    sv = (C.c_float * len(svb))(*[float(v) for v in svb])
    st = L.rnde_tapes_backward(t, tid, ub.data_ptr(), sv, xb.data_ptr(), pb.data_ptr(), tsb, None)
    return st, xb.cpu().numpy(), pb.cpu().numpy()


def test_two_tapes_and_a_probe_in_between():
    from regneuralde_jl_amd import _lib
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    L = _lib.lib()
    arch, p, x = _setup("small", 24, 31, 4.0)
    x2 = np.random.default_rng(32).uniform(-1, 1, x.shape).astype(np.float32)
    cfg = _cfg(arch, 24, reltol=1e-3, abstol=1e-3, col_tile=16)
    t = C.c_void_p()
    assert L.rnde_tapes_create(C.byref(cfg), 2, C.byref(t)) == 0, L.rnde_tapes_last_error(None)
    P = L.rnde_param_count(C.byref(cfg))
    st, u1, nfe1, sv1, id1 = _fwd(L, t, x, p, 1)
    assert st == 0 and id1 == 0
    st, up, nfep, _, idp = _fwd(L, t, x2, p, 0)                     # the probe: untaped, must not disturb tape 0
    assert st == 0 and idp == -1 and L.rnde_tapes_in_use(t) == 1
    st, u2, nfe2, sv2, id2 = _fwd(L, t, x2, p, 1)                    # a second batch in flight
    assert st == 0 and id2 == 1 and nfe2 == nfep and np.array_equal(u2, up)
    st, _, _, _, id3 = _fwd(L, t, x, p, 1)                           # pool exhausted: an error, never a silently dropped tape
    assert st == _lib.BAD_ARG and id3 == -1 and b"in use" in L.rnde_tapes_last_error(t)
    ubar = np.random.default_rng(33).standard_normal(x.shape).astype(np.float32)
    # reverse in the order of the forwards' ids reversed; each must equal a plain handle's forward + reverse of the same batch
    for tid, xx, sv in ((id2, x2, sv2), (id1, x, sv1)):
        st, xb, pb = _bwd(L, t, tid, ubar, np.full(len(sv), 2.0), P)
        assert st == 0
        n = Node(cfg)
        n.forward(xx, p, keep_tape=True)
        rx, rp, _ = n.backward(ubar, np.full(len(sv), 2.0, dtype=np.float32))
        assert np.array_equal(xb, rx) and np.array_equal(pb, rp)
    assert L.rnde_tapes_in_use(t) == 0
    st, _, _ = _bwd(L, t, id1, ubar, np.full(len(sv1), 2.0), P)       # a tape is consumed by its reverse pass
    assert st == _lib.NO_TAPE
    st, _, _, _, id4 = _fwd(L, t, x, p, 1)
    assert st == 0 and id4 == 0 and L.rnde_tapes_release(t, id4) == 0 and L.rnde_tapes_in_use(t) == 0
    assert L.rnde_node_fallback_count(L.rnde_tapes_node(t, 0)) == 0
    L.rnde_tapes_destroy(t)
