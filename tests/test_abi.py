"""The C-ABI library loads and exports every symbol include/rnde.h declares (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rnde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rnde_[a-z_]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ["rnde_node_create", "rnde_node_forward", "rnde_node_backward", "rnde_node_release_tape", "rnde_node_destroy",
                 "rnde_last_error", "rnde_version"]:
        assert must in names


def test_library_exports_every_declared_symbol(rnde):
    L = rnde._lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert set(rnde._lib.EXPORTS) == set(_declared())
    assert b"gfx950" in L.rnde_version()


def test_param_count_matches_flux_layout(rnde):
    cfg = rnde._lib.NodeConfig()
    cfg.n_layers = 2
    cfg.dims[0], cfg.dims[1], cfg.dims[2] = 784, 100, 784
    cfg.time_dep = 1
    assert rnde._lib.lib().rnde_param_count(C.byref(cfg)) == 158568
    cfg.dims[0], cfg.dims[1], cfg.dims[2] = 2, 10, 2
    assert rnde._lib.lib().rnde_param_count(C.byref(cfg)) == 3 * 10 + 10 + 11 * 2 + 2


def test_create_without_gpu_fails_loudly(rnde):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = rnde._lib.NodeConfig()
    cfg.n_layers = 2
    cfg.dims[0], cfg.dims[1], cfg.dims[2] = 2, 10, 2
    cfg.act[0] = 1
    cfg.time_dep = 1
    cfg.max_batch, cfg.max_attempts, cfg.reltol, cfg.abstol = 4, 8, 1e-3, 1e-3
    h = C.c_void_p()
    st = rnde._lib.lib().rnde_node_create(C.byref(cfg), C.byref(h))
    assert st == rnde._lib.NO_DEVICE and not h.value
    assert b"device" in rnde._lib.lib().rnde_last_error(None)


def test_product_package_never_imports_the_oracle():
    pk = os.path.join(ROOT, "regneuralde.jl_amd")
    for dp, _, fns in os.walk(pk):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "rnde_oracle" not in txt, fn
