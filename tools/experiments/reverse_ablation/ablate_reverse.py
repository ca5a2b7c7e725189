"""Ablation of the REVERSE attempt kernel (VERDICT r04 "next" item 3): the same launch geometry, the attempt sequence fixed by the (untouched) forward
solve's step log, with (i) polls replaced by immediate reads, (ii) no tape DMA in START, (iii) no tape stores, (iv) no cross-workgroup partials /
scalar chain, (v) the MFMAs of the six stages alone, (vi) START alone, (vii) no END reduction.  Variants are built by build_variants.sh beside this
file (patched COPIES of rnde_bstage_persist.h / rnde_stage_persist.h; their RESULTS are wrong by construction).  Each variant runs in its own process:
8 training-step gradients at fixed weights, HIP events of the library around the reverse sweep's attempt launches, min over 3 rounds.
    python tools/experiments/reverse_ablation/ablate_reverse.py > profiles/r05_rev_attempt_ablation.csv"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
code = r'''
import ctypes as C, sys, os, torch
sys.path.insert(0, '.')
import bench, regneuralde_jl_amd as rn
from regneuralde_jl_amd import _lib
L = _lib.lib()
B = int(os.environ['AB_B'])
dev = torch.device("cuda", 0)
model = bench.build_model(rn, dev, B)
g = torch.Generator().manual_seed(1999)
x = torch.rand(B, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].to(dev)
rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
h = model.node._acquire(x.reshape(B, -1), True)
L.rnde_node_set_timing(h.ptr, 1)
fa = rs = rr = n = 0
reps = 8 if B <= 512 else 3
for _ in range(reps):
    rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    L.rnde_node_timing(h.ptr, C.byref(a), C.byref(b), C.byref(c))
    fa += a.value; rs += b.value; rr += c.value; n += int(L.rnde_node_last_attempts(h.ptr))
print("%.2f %.2f %d" % (1e3 * rs / n, 1e3 * fa / n, n // reps))
'''
variants = [("base", "the shipped kernel"), ("nopoll", "(i) polls replaced by immediate reads"), ("nodma", "(ii) START does not bring the stages' tape operands into LDS"),
            ("nostore", "(iii) no tape stores (k-bar in place, z1-bar)"), ("noscalar", "(iv) no cross-workgroup partials, no f64 scalar chain in START"),
            ("nopoll_nodma_nostore", "(i) + (ii) + (iii)"),
            ("mfmaonly", "(v) the 6 x (25 + 28) MFMAs per wave alone, operands in registers (+ prologue and the whole of START)"),
            ("startonly", "(vi) launch + prologue + START alone (the kernel ends behind START's phase D)"),
            ("startonly_noscalar", "(vi) without (iv)'s partials and scalar chain"),
            ("noend", "(vii) no END (the reduction of the 21 partial sums)")]
print("# reverse attempt kernel rnde_bstage_attempt_kernel<1,1>, in-sweep launches (HIP events of the library), us per reversed attempt (min of 3 rounds); MI355X")
print("variant,what,B,us_rev_attempt,us_fwd_attempt_same_run,attempts")
for B in (512, 4096):
    for name, what in variants:
        lib = os.path.join(ROOT, "regneuralde.jl_amd", "lib", f"librnde_rabl_{name}.so")
        if not os.path.exists(lib):
            continue
        best = [1e9, 1e9, 0]
        for rep in range(3):
            out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RNDE_LIB=lib, AB_B=str(B)), capture_output=True, text=True, cwd=ROOT)
            try:
                a, b, c = out.stdout.split()
                best = [min(best[0], float(a)), min(best[1], float(b)), int(c)]
            except Exception:
                sys.stderr.write(out.stderr[-400:])
        print(f'{name},"{what}",{B},{best[0]:.2f},{best[1]:.2f},{best[2]}', flush=True)
