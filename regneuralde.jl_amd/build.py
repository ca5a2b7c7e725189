"""Build librnde.so (the C-ABI HIP library) in-tree with hipcc for gfx950: one object per translation unit (in parallel), then link."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# RNDE_LIB: load another build of the same library (A/B runs of kernel variants on one GPU box; tools/ab_bench.sh)
LIB = os.environ.get("RNDE_LIB") or os.path.join(_HERE, "lib", "librnde.so")
SOURCES = ["rnde.hip", "rnde_reverse.hip", "rnde_stage_solve.hip", "rnde_latent.hip", "rnde_sde.hip", "rnde_comm.hip", "rnde_tapes.hip"]


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(_HERE, "..", "include", "rnde.h")]


def _flag_stamp():
    """The compile-time switches a build was made with (a variant left in place by an A/B run must not pass for the default build)."""
    return "RNDE_EXTRA_FLAGS=%s" % os.environ.get("RNDE_EXTRA_FLAGS", "")


def needs_build():
    if not os.path.exists(LIB):
        return True
    stamp = LIB + ".flags"
    if os.path.exists(stamp) and open(stamp).read() != _flag_stamp():      # (no stamp: a prebuilt library that travelled without it)
        return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > m for f in [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-undefined-internal"]
    flags += os.environ.get("RNDE_EXTRA_FLAGS", "").split()      # compile-time A/B switches (tools/ab_build.sh)
    import re

    def deps_mtime(path, seen=None):
        """newest modification time of a source and of the project headers it includes, recursively (a kernel header that only one
        translation unit includes must not cost a rebuild of the 2-minute monolith)"""
        seen = set() if seen is None else seen
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return 0.0
        seen.add(path)
        m = os.path.getmtime(path)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
            m = max(m, deps_mtime(os.path.join(os.path.dirname(path), inc), seen))
        return m
    restamp = os.path.exists(LIB + ".flags") and open(LIB + ".flags").read() != _flag_stamp()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and not restamp and os.path.exists(obj) and os.path.getmtime(obj) >= deps_mtime(srcp):
            return obj
        cmd = [hipcc] + flags + ["-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(LIB + ".flags", "w") as f:
        f.write(_flag_stamp())
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--incremental" not in sys.argv, verbose=True))      # default: rebuild everything; --incremental: only what changed
