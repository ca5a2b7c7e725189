"""Forward attempt kernel, back to back: untaped / taped into ONE record (warm: it stays in the Infinity Cache) / taped into 2..48 records in turn
(cold: what a solve does).  python tools/cold_tape.py"""
import ctypes as C, sys, os
sys.path.insert(0, '.')
from tests.test_gpu_forward import _setup, _cfg
from tests.util import Node
B = int(os.environ.get('AB_B', '512'))
arch, p, x = _setup("mnist", B, 7, 1.0)
n = Node(_cfg(arch, B, max_attempts=64, col_tile=16))
us = C.c_float(0)
xd, pd = n.dev(x), n.dev(p)
for rep in range(2):
    n.L.rnde_bench_attempt(n.h, xd.data_ptr(), pd.data_ptr(), B, 300, C.byref(us), None); a = us.value
    n.L.rnde_bench_attempt_taped(n.h, xd.data_ptr(), pd.data_ptr(), B, 300, C.byref(us), None); b = us.value
    out = []
    for recs in (2, 4, 8, 16, 32, 48):
        st = n.L.rnde_bench_attempt_cold_tape(n.h, xd.data_ptr(), pd.data_ptr(), B, 300, recs, C.byref(us), None)
        out.append("%d: %.2f" % (recs, us.value) if st == 0 else "%d: status %d" % (recs, st))
    print("untaped %.2f  one record %.2f  records in turn  %s" % (a, b, "  ".join(out)), flush=True)
