"""Device vs the oracle's device-order summation mode (oracle sum_order = 3): one f evaluation, natural runs at the reference
tolerance (B = 64 x 16 seeds, B = 512 x 2 seeds), replay along the mode-3 oracle's sequence.  Diagnostic behind
tests/test_gpu_replay.py (round 3); run on the GPU box:  python tools/devorder_check.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import Node, Oracle, arch_mnist, glorot_params, make_cfg

TOL = 1.4e-8


def prob(B, seed):
    rng = np.random.default_rng(seed)
    arch = arch_mnist()
    return arch, glorot_params(arch, rng, np.float32, 1.0), rng.uniform(0, 1, (B, 784)).astype(np.float32)


def main():
    arch, p, x = prob(64, 3)
    node = Node(make_cfg([784, 100, 784], ["tanh", "tanh"], 64, reltol=TOL, abstol=TOL, max_attempts=96))
    fd = node.feval(x, p, 0.37)
    for mode in (0, 1, 3):
        fo = Oracle(arch, np.float32, TOL, TOL, sum_order=mode).f_eval(p, x, 0.37)
        print(f"f eval, oracle mode {mode}: equal {np.mean(fd == fo):.4f}, max |diff| {np.abs(fd - fo).max():.2e}, rms {np.sqrt(np.mean((fd - fo) ** 2)):.2e}")
    for B, seeds in ((64, range(100, 116)), (512, (11, 12))):
        node = Node(make_cfg([784, 100, 784], ["tanh", "tanh"], B, reltol=TOL, abstol=TOL, max_attempts=96))
        for seed in seeds:
            arch, p, x = prob(B, seed)
            o3 = Oracle(arch, np.float32, TOL, TOL, reg_kind=1, max_attempts=96, sum_order=3)
            r3 = o3.forward(x, p)
            se = o3.steps_ext()
            g = node.forward(x, p)
            n = min(g["nattempts"], r3["nattempts"])
            dtd, dto = g["steps"][:n, 1], se[:n, 1]
            ee = g["steps"][:n, 2] / se[:n, 3]
            print(f"B={B} seed {seed}: attempts device {g['nattempts']} oracle3 {r3['nattempts']}; dt rel diff max {np.abs(dtd / dto - 1).max():.2e}; "
                  f"EEst ratio {ee.min():.3f}..{ee.max():.3f} mean {ee.mean():.3f}; u diff {np.abs(g['u'] - r3['u']).max():.2e}; sv sum {g['saveval'].sum():.5f} / {r3['saveval'].sum():.5f}")
            if seed == seeds[0]:
                rep = node.forward_replay(x, p, se[:, 2], se[:, 4])
                er = rep["steps"][:, 2] / se[:, 3]
                print(f"   replay along oracle3: EEst ratio {er.min():.3f}..{er.max():.3f} mean {er.mean():.3f}; u equal frac {np.mean(rep['u'] == r3['u']):.4f} max diff {np.abs(rep['u'] - r3['u']).max():.2e}")
        node.close()


if __name__ == "__main__":
    main()
