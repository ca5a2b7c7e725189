"""Compressed view of a kernel's instruction stream from device assembly (tools/kernel_regs.sh leaves it in /tmp): one character per instruction
(M matrix, v vector, r/w LDS read/write, L/S global load/store, | waitcnt, B barrier, s scalar, j branch; labels start a new line).
    python tools/isa_mix.py /tmp/rnde_reverse_.s rnde_wgrad4x_kernelILb1"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
start = None
lines = txt.split("\n")
for i, l in enumerate(lines):
    if pat in l and not l.startswith("\t") and not l.startswith(".") and ":" in l.split(";")[0]:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or ".Lfunc_end" in lines[i])


def cls(l):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        return "\n" if l.endswith(":") and l.startswith(".LBB") else None
    op = l.split()[0]
    for pre, c in (("v_mfma", "M"), ("ds_read", "r"), ("ds_write", "w"), ("buffer_load", "L"), ("global_load", "L"), ("buffer_store", "S"), ("global_store", "S"),
                   ("s_waitcnt", "|"), ("s_barrier", "B"), ("s_cbranch", "j"), ("s_branch", "j"), ("s_", "s"), ("v_", "v")):
        if op.startswith(pre):
            return c
    return "?"


print("".join(c for c in (cls(l) for l in lines[start:end]) if c))
