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


def _c_layout(tmp_path):
    """sizeof / offsetof of the two config structs as gcc lays them out (tests/abi_c/abi_check.c against include/rnde.h)."""
    import subprocess
    exe = os.path.join(str(tmp_path), "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_c", "abi_check.c"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    lay, size = {}, {}
    for line in out.splitlines():
        w = line.split()
        if w[0] == "constants":
            size["constants"] = dict(zip(w[1::2], map(int, w[2::2])))
        elif w[1] == "sizeof":
            size[w[0]] = int(w[2])
        else:
            lay.setdefault(w[0], []).append((w[1], int(w[2]), int(w[3])))
    return lay, size


def _julia_struct(name):
    """(field, type) list of `struct <name> ... end` in bindings/julia/RNDE.jl (one or several `a::T; b::T` declarations per line)."""
    src = open(os.path.join(ROOT, "bindings", "julia", "RNDE.jl")).read()
    body = re.search(r"\nstruct " + name + r"\b[^\n]*\n(.*?)\nend\n", src, flags=re.S).group(1)
    fields = []
    for line in body.splitlines():
        line = line.split("#")[0]
        for decl in line.split(";"):
            m = re.match(r"\s*(\w+)::(.+?)\s*$", decl)
            if m:
                fields.append((m.group(1), m.group(2)))
    return fields


_JULIA_BYTES = {"Int32": 4, "Float32": 4, "NTuple{9,Int32}": 36, "NTuple{8,Int32}": 32}


@pytest.mark.parametrize("cname,pyname,jlname", [("rnde_node_config", "NodeConfig", "NodeConfig"), ("rnde_nsde_config", "NsdeConfig", "NsdeConfig")])
def test_config_struct_layouts_agree_between_c_ctypes_and_julia(rnde, tmp_path, cname, pyname, jlname):
    """One layout, three descriptions: the C header (compiled), the ctypes Structure the Python host and every GPU test use, and the Julia
    struct a maintainer's `ccall` passes by reference -- same fields in the same order at the same offsets with the same widths."""
    lay, size = _c_layout(tmp_path)
    c_fields = lay[cname]
    py = getattr(rnde._lib, pyname)
    assert C.sizeof(py) == size[cname]
    assert [f[0] for f in py._fields_] == [f[0] for f in c_fields]
    for (fname, off, nbytes) in c_fields:
        d = getattr(py, fname)
        assert (d.offset, d.size) == (off, nbytes), fname
    jl = _julia_struct(jlname)
    assert [f[0] for f in jl] == [f[0] for f in c_fields]
    off = 0
    for (fname, ty), (_, coff, cbytes) in zip(jl, c_fields):
        assert ty in _JULIA_BYTES, (fname, ty)
        assert (off, _JULIA_BYTES[ty]) == (coff, cbytes), fname      # (all members are 4-byte aligned: Julia lays an isbits struct out as C does)
        off += _JULIA_BYTES[ty]
    assert off == size[cname]
    assert size["constants"] == {"RNDE_MAX_LAYERS": 8, "RNDE_COMM_ID_BYTES": 128, "RNDE_COMM_WINDOW_BYTES": 64}


def test_julia_binding_calls_only_declared_symbols_and_binds_the_module_name():
    """bindings/julia/*.jl cannot run here (no Julia): what can be checked is that every `ccall` names an entry point the header declares,
    that the module name AMDGPU is bound before AMDGPU.stream() / AMDGPU.synchronize() are used (round-3 review: `using AMDGPU: ...` alone
    does not bind it), that a Tracker rule exists for each of the four solve shapes, and that the patch files define all eight call methods."""
    d = os.path.join(ROOT, "bindings", "julia")
    src = open(os.path.join(d, "RNDE.jl")).read()
    declared = set(_declared())
    called = set(re.findall(r"ccall\(\(:(\w+), LIB\)", src))
    assert called and called <= declared, called - declared
    assert re.search(r"^import AMDGPU\b", src, flags=re.M) and src.index("import AMDGPU") < src.index("AMDGPU.stream()")
    for rule in ("rnde_solve", "rnde_solve_saveat", "rnde_nsde_solve", "rnde_nsde_solve_saveat"):
        assert re.search(r"@grad function " + rule + r"\(", src), rule
    ode = open(os.path.join(d, "patch_neural_ode.jl")).read()
    sde = open(os.path.join(d, "patch_neural_sde.jl")).read()
    for r in ("false,false", "false,true", "true,false", "true,true"):
        assert re.search(r"^function \(n::TrackedNeuralODE\{" + r + r"\}\)\(x, p = n\.p;", ode, flags=re.M), r
        assert re.search(r"^function \(n::TrackedNeuralDSDE\{" + r + r"\}\)\(x, p = n\.p;", sde, flags=re.M), r
