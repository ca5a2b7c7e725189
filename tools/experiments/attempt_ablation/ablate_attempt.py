"""Ablation of the forward attempt kernel: the same launch geometry with (i) polls replaced by immediate reads,
(ii) tanh replaced by a move, (iii) no tape stores, (iv) the MFMAs alone with operands in registers -- the measured floor per attempted step.
Variants are built by build_variants.sh beside this file (a patched COPY of rnde_stage_persist.h; their RESULTS are wrong by construction).
Each variant runs in its own process; back-to-back forced attempts (rnde_bench_attempt / _taped), B = 512 and B = 4096, three rounds.
    python tools/experiments/attempt_ablation/ablate_attempt.py > profiles/r04_attempt_ablation.csv"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
code = r'''
import ctypes as C, sys, os
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = int(os.environ['AB_B'])
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=16 if B > 1024 else 64, col_tile=16))
us = C.c_float(0); ust = C.c_float(0)
n.L.rnde_bench_attempt(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 300, C.byref(us), None)
n.L.rnde_bench_attempt_taped(n.h, n.dev(x).data_ptr(), n.dev(p).data_ptr(), B, 300, C.byref(ust), None)
print("%.2f %.2f" % (us.value, ust.value))
'''
variants = [("base", "the shipped kernel"), ("nopoll", "(i) polls replaced by immediate reads"), ("notanh", "(ii) tanh replaced by a move"),
            ("notape", "no tape stores"), ("nopoll_notanh", "(i) + (ii)"), ("nopoll_notanh_notape", "(i) + (ii) + no tape stores"),
            ("mfmaonly", "(iii) the 6 x (26 + 28) MFMAs per wave alone, operands in registers (+ prologue: weight loads, controller, START)")]
print("# forward attempt kernel rnde_stage_attempt_kernel<1,1>, back-to-back forced attempts, us per attempted step (min of 3 rounds); MI355X")
print("variant,what,B,us_untaped,us_taped")
for B in (512, 4096):
    for name, what in variants:
        lib = os.path.join(ROOT, "regneuralde.jl_amd", "lib", f"librnde_abl_{name}.so")
        if not os.path.exists(lib):
            continue
        best = [1e9, 1e9]
        for rep in range(3):
            out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RNDE_LIB=lib, AB_B=str(B), RNDE_PERSIST2="0"), capture_output=True, text=True, cwd=ROOT)
            try:
                a, b = map(float, out.stdout.split())
                best = [min(best[0], a), min(best[1], b)]
            except Exception:
                sys.stderr.write(out.stderr[-400:])
        print(f'{name},"{what}",{B},{best[0]:.2f},{best[1]:.2f}', flush=True)
