"""How the CPU restatement (bench.py cpu_baseline) scales with OpenMP threads on the GPU box's host (round 3: 256 threads measured
1.95 samples/s against 75 on ONE thread -- oversubscription).  python tools/cpu_threads_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle, arch_mnist, glorot_params
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "OMP env", {k: v for k, v in os.environ.items() if "OMP" in k or "GOMP" in k})
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
arch = arch_mnist()
rng = np.random.default_rng(1999)
p = glorot_params(arch, rng)
x = rng.uniform(0, 1, (512, 784)).astype(np.float32)
ub = (rng.standard_normal((512, 784)) / 512).astype(np.float32)
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > len(os.sched_getaffinity(0)):
        break
    nb = 64 if th == 1 else 512
    orc = Oracle(arch, np.float32, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1, max_attempts=400)
    orc.lib.orc_set_threads(ctypes.c_int(th))
    t0 = time.perf_counter(); r = orc.forward(x[:nb], p); t1 = time.perf_counter()
    orc.backward(ub[:nb], np.full(len(r["saveval"]), 1.0, np.float32)); t2 = time.perf_counter()
    print(f"threads {th:3d} batch {nb}: forward {t1 - t0:.2f} s, backward {t2 - t1:.2f} s -> {nb / (t2 - t0):.1f} samples/s", flush=True)
    if t2 - t0 > 60:
        break
