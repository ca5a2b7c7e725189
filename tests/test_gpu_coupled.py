"""SURVEY 8e mode 2 -- one step-size controller for all shards (rnde_node_set_coupling) -- on the one GPU of the test box.

world = 2 and 4 inside ONE process: one handle, one host thread and one stream per rank, the in-process communicator (rnde_comm_create_local_group: the
all-reduce is a one-workgroup kernel per rank meeting the other through device memory).  The sharded run must reproduce the
single-device run over the whole batch: the same accept/reject sequence, step sizes and saved values to fp32 rounding (the error
norm is now a sum of two partial sums), the end state of every column, and -- with the ranks' cotangents formed the way data
parallelism forms them (data term = mean over the rank's own columns) -- gradients whose AVERAGE over the ranks is the
single-device gradient.  world = 1 through RCCL: the collective's plumbing, bit-identical to the uncoupled run."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(B, seed, scale, kind="small"):
    if kind == "small":
        from tests.test_gpu_forward import _setup as base
        return base("small", B, seed, scale)      # MNIST-form network (D = 36, H = 10) on the stage engine
    from tests.test_gpu_chain import _setup as chain
    return chain(kind, B, seed, scale)            # Dense chains on the chain engine's multi-wave kernels (col_tile 65)


def _run(node, x, p, ubar, svb):
    g = node.forward(x, p, keep_tape=True)
    gx, gp, gt = node.backward(ubar, np.full(len(g["saveval"]), svb, dtype=np.float32))
    return g, gx, gp, gt


@pytest.mark.parametrize("persist", [1, 0])
@pytest.mark.parametrize("B,world,tol,scale", [(64, 2, 1e-3, 5.0), (70, 2, 1e-4, 4.0), (128, 4, 1e-4, 4.0)])
def test_coupled_shards_reproduce_the_single_device_run(B, world, tol, scale, persist, monkeypatch):
    monkeypatch.setenv("RNDE_PERSIST", str(persist))
    _coupled_case("small", 16, B, world, tol, scale)


@pytest.mark.parametrize("kind,B,world,tol,scale", [("latent", 64, 2, 1e-4, 2.0), ("latent", 70, 2, 1e-3, 2.0), ("chain3", 128, 4, 1e-4, 2.0),
                                                      ("wide", 96, 2, 1e-3, 1.5)])
@pytest.mark.parametrize("svb,gtol", [(0.0, 2e-4), (3.0, 1e-3)])
def test_coupled_shards_on_the_chain_engine(kind, B, world, tol, scale, svb, gtol):
    """The same property on the chain engine's multi-wave kernels (one launch per attempt when coupled: the one-launch solve keeps its
    controller inside the kernel and is not used with a shared controller).  Without a cotangent on the saved EEst*dt values the
    gradients agree as on the stage engine; with one, the term's own fp32 noise (the gradient of a rounding-floor quantity)
    sets the bound."""
    _coupled_case(kind, 65, B, world, tol, scale, svb, gtol)


def _coupled_case(kind, tile, B, world, tol, scale, svb=3.0, gtol=2e-4):
    from regneuralde_jl_amd import _lib
    from tests.test_gpu_forward import _cfg as _cfg0
    from tests.util import Node
    _cfg = lambda a, b, **kw: _cfg0(a, b, **{**kw, "col_tile": tile})
    L = _lib.lib()
    arch, p, x = _setup(B, 21, scale, kind)
    rng = np.random.default_rng(22)
    ubar = rng.standard_normal(x.shape).astype(np.float32) / B          # data term of the single-device loss: a mean over all columns
    ref, rx, rp, rt = _run(Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16)), x, p, ubar, svb)
    print("attempts", len(ref["steps"]), "rejected", int((ref["steps"][:, 3] == 0).sum()))

    comms = (C.c_void_p * world)()
    assert L.rnde_comm_create_local_group(world, 0, comms) == 0, L.rnde_comm_last_error(None)
    per = B // world                      # equal shards, as data parallelism makes them (every rank must hold the same number of 16-column tiles)
    assert per * world == B
    shards = [(r * per, (r + 1) * per) for r in range(world)]
    nodes = [Node(_cfg(arch, per, reltol=tol, abstol=tol, col_tile=16)).own_stream() for _ in shards]
    for n, c in zip(nodes, comms):
        _lib.check(n.h, L.rnde_node_set_coupling(n.h, C.c_void_p(c), B))
    out = [None] * world

    def work(r):
        lo, hi = shards[r]
        # what a data-parallel rank passes: the cotangent of ITS loss, whose data term is a mean over its own columns (= world x the
        # single-device cotangent of those columns)
        out[r] = _run(nodes[r], x[lo:hi], p, ubar[lo:hi] * float(world), svb)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert all(o is not None for o in out), "a rank did not finish"
    for r, (lo, hi) in enumerate(shards):
        g = out[r][0]
        assert g["nfe"] == ref["nfe"] and np.array_equal(g["steps"][:, 3], ref["steps"][:, 3])
        np.testing.assert_allclose(g["steps"][:, 1], ref["steps"][:, 1], rtol=2e-5)
        # (EEst*dt: the error estimate is a difference of stage values, so last-bit differences of dt show at ~1e-3 relative on the
        #  three-layer chains at tol 1e-4, where it sits closer to its fp32 rounding floor than on the MNIST form)
        np.testing.assert_allclose(g["saveval"], ref["saveval"], rtol=2e-4 if kind == "small" else 2e-3, atol=1e-9)
        assert np.abs(g["u"] - ref["u"][lo:hi]).max() <= 2e-5 * max(1.0, np.abs(ref["u"]).max())
        # every rank holds the same controller history bit for bit
        assert np.array_equal(g["steps"], out[0][0]["steps"]) and np.array_equal(g["saveval"], out[0][0]["saveval"])
    # the usual average of the ranks' gradients is the single-device gradient
    gp = sum(o[2] for o in out) / world
    gt = sum(o[3] for o in out) / world
    scale_p = np.abs(rp).max()
    print("p-bar", np.abs(gp - rp).max() / scale_p, "tspan-bar", gt, rt)
    assert np.abs(gp - rp).max() <= gtol * scale_p
    assert np.abs(gt - rt).max() <= gtol * max(1.0, np.abs(rt).max())
    for r, (lo, hi) in enumerate(shards):
        assert np.abs(out[r][1] / world - rx[lo:hi]).max() <= gtol * np.abs(rx).max()
    for c in comms:
        L.rnde_comm_destroy(C.c_void_p(c))


def test_coupling_with_one_rank_through_rccl_is_the_uncoupled_run():
    from regneuralde_jl_amd import _lib
    from tests.test_gpu_forward import _cfg
    from tests.util import Node
    L = _lib.lib()
    arch, p, x = _setup(48, 23, 3.0)
    ubar = np.random.default_rng(24).standard_normal(x.shape).astype(np.float32)
    a = _run(Node(_cfg(arch, 48, reltol=1e-3, abstol=1e-3, col_tile=16)), x, p, ubar, 2.0)
    buf = C.create_string_buffer(128)
    assert L.rnde_comm_unique_id(buf) == 0
    comm = C.c_void_p()
    assert L.rnde_comm_create(bytes(buf.raw), 0, 1, 0, C.byref(comm)) == 0
    n = Node(_cfg(arch, 48, reltol=1e-3, abstol=1e-3, col_tile=16))
    _lib.check(n.h, L.rnde_node_set_coupling(n.h, comm, 48))
    b = _run(n, x, p, ubar, 2.0)
    assert a[0]["nfe"] == b[0]["nfe"] and np.array_equal(a[0]["u"], b[0]["u"]) and np.array_equal(a[0]["saveval"], b[0]["saveval"])
    for u, v in zip(a[1:], b[1:]):
        assert np.array_equal(u, v)
    _lib.check(n.h, L.rnde_node_set_coupling(n.h, None, 0))
    n.close()
    L.rnde_comm_destroy(comm)


@pytest.mark.parametrize("tile", [64])
def test_coupling_is_refused_on_the_one_wave_kernels(tile):
    """The shared controller lives where the headline paths live: the stage engine and the chain engine's multi-wave kernels."""
    from regneuralde_jl_amd import _lib
    from oracle.oracle import arch_latent
    from tests.test_gpu_forward import _cfg
    from tests.util import Node
    L = _lib.lib()
    comms = (C.c_void_p * 1)()
    assert L.rnde_comm_create_local_group(1, 0, comms) == 0
    n = Node(_cfg(arch_latent(), 16, col_tile=tile))
    assert L.rnde_node_set_coupling(n.h, C.c_void_p(comms[0]), 16) == _lib.BAD_ARG
    L.rnde_comm_destroy(C.c_void_p(comms[0]))
