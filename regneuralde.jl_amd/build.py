"""Build librnde.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# RNDE_LIB: load another build of the same library (A/B runs of kernel variants on one GPU box; tools/ab_bench.sh)
LIB = os.environ.get("RNDE_LIB") or os.path.join(_HERE, "lib", "librnde.so")
SOURCES = ["rnde.hip"]
HEADERS = ["rnde_device.h", "rnde_fwd.h", "rnde_bwd.h", "rnde_stage.h", "rnde_bstage.h", "rnde_stage_persist.h", "rnde_bstage_persist.h",
           "rnde_chain.h", "rnde_quad.h", "rnde_bchain.h", "rnde_head.h", os.path.join("..", "..", "include", "rnde.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > m for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
