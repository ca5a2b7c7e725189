"""GPU: edge cases and error behaviour of the C ABI (status codes of include/rnde.h), plus a full-size replication
property (size-independent properties at BASELINE.json's sizes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, BAD_ARG, MAX_ATT, DT_UNDER, NONFINITE, HIP, NO_TAPE = 0, 1, 2, 3, 4, 5, 6


def _mk(kind="small", B=9, **kw):
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    arch, p, x = _setup(kind, B, 3, kw.pop("scale", 3.0))
    return Node(_cfg(arch, kw.pop("max_batch", B), **kw)), p, x


def test_max_attempts_is_reported_not_hidden():
    """The reference never checks retcode (SURVEY 8b 'Errors'); the ABI returns RNDE_ERR_MAX_ATTEMPTS."""
    from regneuralde_jl_amd._lib import RndeError
    node, p, x = _mk("small", 9, reltol=1e-6, abstol=1e-6, max_attempts=3, col_tile=16)
    with pytest.raises(RndeError) as e:
        node.forward(x, p)
    assert e.value.status == MAX_ATT
    node2, p, x = _mk("small", 9, reltol=1e-6, abstol=1e-6, max_attempts=3, col_tile=64)     # chain engine
    with pytest.raises(RndeError) as e:
        node2.forward(x, p)
    assert e.value.status == MAX_ATT


@pytest.mark.parametrize("col_tile", [16, 64])
def test_nonfinite_input_is_reported(col_tile):
    from regneuralde_jl_amd._lib import RndeError
    node, p, x = _mk("small", 9, reltol=1e-3, abstol=1e-3, col_tile=col_tile)
    x = x.copy(); x[4, 7] = np.nan
    with pytest.raises(RndeError) as e:
        node.forward(x, p)
    assert e.value.status == NONFINITE


def test_bad_arguments():
    from regneuralde_jl_amd._lib import RndeError
    from tests.util import Node, make_arch
    from tests.test_gpu_forward import _cfg
    node, p, x = _mk("small", 9, reltol=1e-3, abstol=1e-3, col_tile=16, max_batch=9)
    with pytest.raises(RndeError) as e:                      # batch larger than the handle was created for
        node.forward(np.concatenate([x, x]), p)
    assert e.value.status == BAD_ARG
    with pytest.raises(RndeError) as e:                      # tspan reversed
        node.forward(x, p, 1.0, 0.0)
    assert e.value.status == BAD_ARG
    for bad in ([0.5, 0.2], [0.5, 1.5], [-0.1, 0.3]):        # saveat not increasing / outside tspan
        with pytest.raises(RndeError) as e:
            node.forward_saveat(x, p, np.array(bad, dtype=np.float32))
        assert e.value.status == BAD_ARG
    with pytest.raises(RndeError) as e:                      # reverse pass without a recorded forward
        node.backward(np.zeros_like(x), None)
    assert e.value.status == NO_TAPE
    wide = make_arch([8, 100, 100, 8], ["tanh", "tanh", "identity"], False)   # not the MNIST form and wider than the chain engine
    with pytest.raises(RndeError) as e:
        Node(_cfg(wide, 4))
    assert e.value.status == BAD_ARG


@pytest.mark.parametrize("B", [1, 15, 17])
def test_ragged_batches_match_the_padded_run(B):
    """Columns are independent given the step sequence: a batch that is not a multiple of the 16-column tile gives the same
    states as the same columns inside a larger batch whose extra columns are copies (same global error norm up to the mean)."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    arch, p, x = _setup("small", B, 4, 3.0)
    a = Node(_cfg(arch, B, reltol=1e-3, abstol=1e-3, col_tile=16)).forward(x, p)
    reps = 16 // B + 2
    xb = np.tile(x, (reps, 1))                               # every column replicated: identical RMS norms
    b = Node(_cfg(arch, B * reps, reltol=1e-3, abstol=1e-3, col_tile=16)).forward(xb, p)
    assert a["nfe"] == b["nfe"]
    np.testing.assert_allclose(b["u"][:B], a["u"], rtol=2e-5, atol=2e-6)
    for r in range(1, reps):
        assert np.array_equal(b["u"][r * B:(r + 1) * B], b["u"][:B])      # replicas are bit-identical


def test_full_size_replication_property():
    """B = 4096 on one GPU (the global batch of SURVEY 8d config 3): 8 copies of 512 distinct MNIST-sized columns.  Each copy
    must come out bit-identical (same arithmetic per column wherever it sits: 256 column tiles, 1792 workgroups), NFE = 3 mod 6,
    and the result must agree with the B = 512 solve of the same columns (same RMS error norm up to summation order)."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    arch, p, x = _setup("mnist", 512, 6, 2.0)
    a = Node(_cfg(arch, 512, reltol=1e-4, abstol=1e-4, col_tile=16)).forward(x, p)
    xb = np.tile(x, (8, 1))
    node = Node(_cfg(arch, 4096, reltol=1e-4, abstol=1e-4, col_tile=16))
    b = node.forward(xb, p, keep_tape=True)
    assert b["nfe"] % 6 == 3 and b["nfe"] == a["nfe"]
    for r in range(1, 8):
        assert np.array_equal(b["u"][r * 512:(r + 1) * 512], b["u"][:512])
    np.testing.assert_allclose(b["u"][:512], a["u"], rtol=1e-4, atol=1e-5)
    ubar = np.tile(np.random.default_rng(0).standard_normal((512, 784)).astype(np.float32), (8, 1))
    gx, gp, gt = node.backward(ubar, np.full(len(b["saveval"]), 2.0, dtype=np.float32))
    for r in range(1, 8):
        assert np.array_equal(gx[r * 512:(r + 1) * 512], gx[:512])
    assert np.isfinite(gp).all() and np.isfinite(gt).all()


def test_large_batch_partial_sums_formed_once_are_bit_identical(monkeypatch):
    """From ~900 per-workgroup partials on (B >= 2048) the reverse sweep sums the partials of attempt n + 1 ONCE behind its launch
    (rnde_bpart_reduce_kernel) instead of in the START of each of the 7 x B / 16 workgroups of attempt n: the same additions in the same order, so
    every cotangent -- including tspan-bar, which is made of exactly those sums -- must be bit-identical to the per-workgroup form
    (RNDE_NO_BPART_REDUCE=1).  B = 2048: 896 partials; rejected steps in the sequence (their cotangents travel through the scalar chain only)."""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    B = 2048
    arch, p, x = _setup("mnist", B, 8, 2.0)
    ubar = np.random.default_rng(3).standard_normal((B, 784)).astype(np.float32) / B
    out = []
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("RNDE_NO_BPART_REDUCE", off)
        else:
            monkeypatch.delenv("RNDE_NO_BPART_REDUCE", raising=False)
        node = Node(_cfg(arch, B, reltol=1e-5, abstol=1e-5, col_tile=16, max_attempts=64))
        f = node.forward(x, p, keep_tape=True)
        g = node.backward(ubar, np.linspace(0.5, 1.5, len(f["saveval"])).astype(np.float32))
        out.append((f, g))
        node.close()
    (fa, ga), (fb, gb) = out
    assert fa["nfe"] == fb["nfe"] and np.array_equal(fa["u"], fb["u"])
    for a, b, name in zip(ga, gb, ("x_bar", "p_bar", "tspan_bar")):
        assert np.array_equal(np.asarray(a), np.asarray(b)), name
    assert np.isfinite(np.asarray(ga[1])).all() and float(np.abs(np.asarray(ga[2])).max()) > 0


def test_persistent_kernel_failure_falls_back_to_the_multi_launch_kernels(monkeypatch):
    """Safety net of rnde_stage_persist.h: when a hand-off gives up (here: a polling bound of zero, so the first poll that is not
    satisfied at once abandons the launch) the handle must notice, switch to the 7-launch kernels and redo the solve --
    the caller sees correct results, not an error and not garbage."""
    import ctypes as C
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    arch, p, x = _setup("mnist", 64, 5, 3.0)
    monkeypatch.setenv("RNDE_PERSIST", "0")
    ref = Node(_cfg(arch, 64, reltol=1e-3, abstol=1e-3, col_tile=16)).forward(x, p)
    monkeypatch.setenv("RNDE_PERSIST", "1")
    monkeypatch.setenv("RNDE_PERSIST_SPINS", "0")
    node = Node(_cfg(arch, 64, reltol=1e-3, abstol=1e-3, col_tile=16))
    node.L.rnde_node_launches_per_attempt.restype = C.c_int32
    assert node.L.rnde_node_launches_per_attempt(node.h) == 1
    got = node.forward(x, p)
    assert node.L.rnde_node_launches_per_attempt(node.h) == 7          # disabled after the failure
    assert got["nfe"] == ref["nfe"] and np.array_equal(got["u"], ref["u"]) and np.array_equal(got["saveval"], ref["saveval"])
    again = node.forward(x, p)
    assert np.array_equal(again["u"], ref["u"])
    # not sticky: after 8 clean multi-launch solves the one-launch kernels get another chance (here they fail again at once,
    # because the polling bound is still zero -- the caller still sees correct results, and the wait doubles)
    node.L.rnde_node_fallback_count.restype = C.c_int32
    assert node.L.rnde_node_fallback_count(node.h) == 1
    for _ in range(6):
        assert np.array_equal(node.forward(x, p)["u"], ref["u"])
    assert node.L.rnde_node_launches_per_attempt(node.h) == 1          # armed again
    assert np.array_equal(node.forward(x, p)["u"], ref["u"])
    assert node.L.rnde_node_fallback_count(node.h) == 2 and node.L.rnde_node_launches_per_attempt(node.h) == 7


@pytest.mark.parametrize("D,H,B", [(8, 50, 21), (20, 15, 9), (40, 31, 33), (130, 47, 18)])
@pytest.mark.parametrize("persist", ["1", "0"])
def test_stage_engine_geometry_corners(D, H, B, persist, monkeypatch):
    """Shapes that exercise the corner paths of the stage kernels: more hidden tiles than waves (D = 8, H = 50: one wave loops over
    4 hidden tiles, each with its own hand-off poll), H + 2 spilling into a further 16-block (H = 15, 31, 47), several row
    blocks with a ragged last tile (D = 130).  Forward + reverse against the fp64 oracle, on both launch paths."""
    from tests.test_gpu_forward import _cfg
    from tests.util import Node, Oracle, arch_mnist, glorot_params, rel_err
    monkeypatch.setenv("RNDE_PERSIST", persist)
    rng = np.random.default_rng(D * 100 + H)
    arch = arch_mnist(D, H)
    p = glorot_params(arch, rng, np.float32, 3.0)
    p = (p + 0.05 * rng.standard_normal(p.shape)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    o64 = Oracle(arch, np.float64, reltol=1e-3, abstol=1e-3, reg_kind=1)
    o32 = Oracle(arch, np.float32, reltol=1e-3, abstol=1e-3, reg_kind=1)
    sa = np.array([0.0, 0.4, 1.0], dtype=np.float32)
    r64, r32 = o64.forward(x, p, saveat=sa), o32.forward(x, p, saveat=sa)
    node = Node(_cfg(arch, B, reltol=1e-3, abstol=1e-3, col_tile=16))
    got = node.forward_saveat(x, p, sa, keep_tape=True)
    assert got["nfe"] == r64["nfe"] == r32["nfe"]
    spread = np.abs(r32["u"] - r64["u"]).max()
    assert np.abs(got["u"] - r64["u"]).max() <= 3e-5 * max(1.0, np.abs(r64["u"]).max()) + 4 * spread
    ubar = rng.standard_normal(r64["u"].shape).astype(np.float32)
    svbar = np.full(len(got["saveval"]), 5.0, dtype=np.float32)
    gx, gp, gt = node.backward(ubar, svbar)
    x64, p64, _ = o64.backward(ubar.astype(np.float64), svbar.astype(np.float64))
    x32, p32, _ = o32.backward(ubar, svbar)
    assert rel_err(gx, x64) <= 2e-3 + 4 * rel_err(x32, x64)
    assert rel_err(gp, p64) <= 2e-3 + 4 * rel_err(p32, p64)


@pytest.mark.parametrize("kind,B,col_tile", [("mnist", 19, 16), ("small", 33, 16), ("test_node", 7, 16), ("mnist", 12, 16)])
def test_results_do_not_depend_on_uninitialised_memory(kind, B, col_tile, monkeypatch):
    """Ragged batches leave padded columns in every tape array.  With RNDE_POISON=1 the library fills each fresh allocation with
    0xFF bytes (NaN): forward and reverse results must be the same, bit for bit, as without it.  (This caught the parameter-gradient
    GEMMs summing over the padded columns -- 0 x whatever the allocator had left there -- which showed up as a NaN gradient in
    the first test run on a fresh GPU box only.)"""
    from tests.test_gpu_forward import _cfg, _setup
    from tests.util import Node
    arch, p, x = _setup(kind, B, 3, 3.0)
    rng = np.random.default_rng(17)
    ubar = rng.standard_normal(x.shape).astype(np.float32)
    out = []
    for poison in (False, True):
        if poison: monkeypatch.setenv("RNDE_POISON", "1")
        else: monkeypatch.delenv("RNDE_POISON", raising=False)
        node = Node(_cfg(arch, B, reltol=1e-3, abstol=1e-3, col_tile=col_tile))
        got = node.forward(x, p, 0.0, 1.0, keep_tape=True)
        gx, gp, gt = node.backward(ubar, np.full(len(got["saveval"]), 5.0, dtype=np.float32))
        assert np.isfinite(got["u"]).all() and np.isfinite(gx).all() and np.isfinite(gp).all() and np.isfinite(gt).all()
        out.append((got["u"], got["saveval"], gx, gp, gt))
    for a, b2 in zip(out[0], out[1]):
        assert np.array_equal(a, b2)
