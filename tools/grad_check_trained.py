"""Is the device's gradient still right after the weights have trained?  Trains the headline model for S steps (bench.py's setup), then compares the
dynamics' gradient of the next step with the CPU oracle's (fp32 and fp64) at the same weights: cosine and norm ratio.
Usage: python tools/grad_check_trained.py S [S2 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import regneuralde_jl_amd as rn
from oracle.oracle import Oracle, arch_mnist
marks = sorted(int(a) for a in sys.argv[1:])
dev = torch.device("cuda:0")
model = bench.build_model(rn, dev, 512)
opt = rn.FluxOptimiser(model.trainable())
g = torch.Generator().manual_seed(1999)
x = torch.rand(512, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (512,), generator=g)].to(dev)
xn, yn = x.reshape(512, -1).cpu().numpy(), y.cpu().numpy()
for i in range(marks[-1] + 1):
    loss, ce, reg, nfe = rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
    if i in marks:
        gd = model.p2.grad.detach().cpu().numpy().astype(np.float64)
        p2 = model.p2.detach().cpu().numpy()
        p3 = model.p3.detach().cpu().numpy()
        W, b = p3[:7840].reshape(784, 10), p3[7840:]
        print(f"step {i}: nfe {nfe} loss {float(loss):.4f} ce {float(ce):.4f} reg {float(reg):.4e} |p2| {np.linalg.norm(p2):.3f} |g_dev| {np.linalg.norm(gd):.4e}", flush=True)
        for dt_ in (np.float32, np.float64):
            orc = Oracle(arch_mnist(784, 100), dt_, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1, max_attempts=400)
            r = orc.forward(xn.astype(dt_), p2.astype(dt_))
            logits = r["u"].astype(np.float64) @ W + b
            z = logits - logits.max(1, keepdims=True)
            sm = np.exp(z) / np.exp(z).sum(1, keepdims=True)
            ce_o = float(-(yn * np.log(sm + 1e-300)).sum(1).mean())
            ubar = ((sm - yn) / 512) @ W.T
            nsv = len(r["saveval"])
            xb, pb, _ = orc.backward(ubar.astype(dt_), np.full(nsv, 100.0 / nsv, dtype=dt_))
            pb = pb.astype(np.float64)
            cos = float(gd @ pb / (np.linalg.norm(gd) * np.linalg.norm(pb)))
            print(f"   oracle {np.dtype(dt_).name}: nfe {r['nfe']} ce {ce_o:.4f} reg {100.0 * float(np.mean(r['saveval'])):.4e} |g| {np.linalg.norm(pb):.4e}  cos(dev, oracle) {cos:.6f}  |dev - oracle| / |oracle| {np.linalg.norm(gd - pb) / np.linalg.norm(pb):.3e}", flush=True)
    opt.step()
