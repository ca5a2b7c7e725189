"""The stage engine's whole adaptive solve as ONE launch (csrc/rnde_stage_solve.h) against the one-launch-per-attempt path and the oracle.

The kernel replaces the body of `solve(prob, Tsit5(); ...)` (reference src/models/neural_ode.jl:131-137) for the MNIST form at <= 512 columns:
attempt loop, PI controller and the cross-workgroup error norm inside the kernel.  It performs the arithmetic of rnde_stage_attempt_kernel
in the same order (per-workgroup partials summed in sum_partials' order), so everything it produces must be bit-identical to that path:
end state, NFE, step log, saved callback values, and -- through the reverse pass that reads it -- the tape.
"""
import ctypes as C

import numpy as np
import pytest

from tests.test_gpu_forward import _cfg, _setup

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_mfma_mode(monkeypatch):
    """This module states properties of the fp32-input-MFMA one-launch solve: bit-identity with the launch-per-attempt kernels (which form their products
    with the same instruction) and with the oracle's device-order mode.  The default matrix mode of new handles (bf16x3 on the matrix cores,
    include/rnde.h: rnde_node_set_matrix_mode) rounds the products differently; its parity is tests/test_gpu_x3.py."""
    monkeypatch.setenv("RNDE_X3", "0")


def _solves(node):
    node.L.rnde_node_one_launch_solves.restype = C.c_int32
    return node.L.rnde_node_one_launch_solves(node.h)


@pytest.mark.parametrize("B,tol,scale,reg", [(512, 1.4e-8, 1.0, 1), (64, 1.4e-8, 1.0, 1), (200, 1e-3, 3.0, 3), (37, 1e-4, 2.0, 2), (512, 1e-3, 4.0, 0)])
def test_one_launch_solve_is_bit_identical_to_launch_per_attempt(B, tol, scale, reg, monkeypatch):
    from tests.util import Node
    arch, p, x = _setup("mnist", B, 5, scale)
    monkeypatch.setenv("RNDE_WGRAD_SIDE", "0")      # (one partition of the weight-gradient GEMMs: p-bar then checks the tape bit for bit)
    outs = []
    for one in ("1", "0"):
        monkeypatch.setenv("RNDE_STAGE_SOLVE", one)
        node = Node(_cfg(arch, B, reltol=tol, abstol=tol, col_tile=16, max_attempts=160, regularize=reg))
        got = node.forward(x, p, keep_tape=True)
        assert _solves(node) == (1 if one == "1" else 0)
        ubar = np.random.default_rng(9).standard_normal(got["u"].shape).astype(np.float32)
        svbar = np.full(len(got["saveval"]), 3.0, dtype=np.float32)
        gx, gp, gt = node.backward(ubar, svbar)
        # a second solve on the same handle (other inputs in between): the meeting's granules carry a new epoch, nothing is stale
        other = node.forward(x[::-1].copy(), p, keep_tape=False)
        again = node.forward(x, p, keep_tape=False)
        assert np.array_equal(again["u"], got["u"]) and again["nfe"] == got["nfe"] and other["nfe"] > 9
        assert node.L.rnde_node_fallback_count(node.h) == 0
        outs.append((got, gx, gp, gt))
        node.close()
    a, b = outs
    assert a[0]["nfe"] == b[0]["nfe"] and a[0]["nfe"] >= 15
    assert np.array_equal(a[0]["u"], b[0]["u"])
    assert np.array_equal(a[0]["saveval"], b[0]["saveval"])
    assert np.array_equal(a[0]["steps"], b[0]["steps"])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


def test_one_launch_solve_matches_the_oracle():
    """End state, NFE, accept / reject pattern and callback values against the CPU oracle (fp32, tol 1e-3: the step sequence is not
    rounding noise there); tolerance 2e-4 on O(1) states, the figure of test_forward_solve_exact_sequence."""
    from tests.util import Node, Oracle
    arch, p, x = _setup("mnist", 96, 11, 3.0)
    ref = Oracle(arch, np.float32, reltol=1e-3, abstol=1e-3, reg_kind=1).forward(x, p)
    node = Node(_cfg(arch, 96, reltol=1e-3, abstol=1e-3, col_tile=16, regularize=1))
    got = node.forward(x, p)
    assert _solves(node) == 1
    assert got["nfe"] == ref["nfe"]
    assert np.abs(got["u"] - ref["u"]).max() <= 2e-4 * max(1.0, np.abs(ref["u"]).max())
    assert len(got["saveval"]) == len(ref["saveval"])
    # (EEst is an O(dt^5) cancellation over the tolerance: 15 % per entry, the tolerance test_forward_solve_exact_sequence states)
    np.testing.assert_allclose(got["saveval"], ref["saveval"], rtol=0.15, atol=3e-6)


def test_one_launch_solve_reports_max_attempts_and_falls_back_on_timeout(monkeypatch):
    """max_attempts ends the launch with the status the per-attempt path reports; a meeting or hand-off that cannot complete (spin bound 0)
    raises the abort word, the solve is redone by the multi-launch kernels and still returns the right answer."""
    from tests.util import Node
    from regneuralde_jl_amd import _lib
    arch, p, x = _setup("mnist", 64, 5, 1.0)
    node = Node(_cfg(arch, 64, reltol=1.4e-8, abstol=1.4e-8, col_tile=16, max_attempts=7))
    with pytest.raises(_lib.RndeError) as ei:
        node.forward(x, p)
    assert "max_attempts" in str(ei.value)
    node.close()
    ref = Node(_cfg(arch, 64, reltol=1e-3, abstol=1e-3, col_tile=16)).forward(x, p)
    monkeypatch.setenv("RNDE_PERSIST_SPINS", "0")
    node = Node(_cfg(arch, 64, reltol=1e-3, abstol=1e-3, col_tile=16))
    got = node.forward(x, p)
    assert node.L.rnde_node_fallback_count(node.h) == 1 and _solves(node) == 0
    assert got["nfe"] == ref["nfe"] and np.array_equal(got["u"], ref["u"])

