"""The headline training run, step by step: NFE and loss every EVERY steps until STEPS or the first error (how fast does the model's NFE grow?).
Usage: python tools/train_trace.py [steps=1500] [every=50]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import regneuralde_jl_amd as rn
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
model = bench.build_model(rn, dev, 512)
opt = rn.FluxOptimiser(model.trainable())
g = torch.Generator().manual_seed(1999)
x = torch.rand(512, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (512,), generator=g)].to(dev)
for i in range(steps):
    try:
        loss, ce, reg, nfe = rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=False)
        opt.step()
    except Exception as e:
        print(f"step {i}: {e}")
        for k, q in enumerate(model.trainable()):
            if q.numel() == 0: continue
            gq = q.grad if q.grad is not None else torch.zeros(1)
            print(f"  param {k}: finite {bool(torch.isfinite(q).all())}  |p| {float(q.norm()):.4e}  max|p| {float(q.abs().max()):.4e}  grad finite {bool(torch.isfinite(gq).all())}  |g| {float(gq.norm()):.4e}")
        for rep in range(3):      # the same weights again: a property of the weights, or of that one solve?
            try:
                loss, ce, reg, nfe = rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
                print(f"  retry {rep}: nfe {nfe} loss {float(loss):.4f}")
            except Exception as e2:
                print(f"  retry {rep}: {e2}")
        break
    if i % every == 0 or i == steps - 1 or (len(sys.argv) > 3 and i >= int(sys.argv[3])):
        print(f"step {i:5d}  nfe {nfe:4d}  loss {float(loss):.4f}  ce {float(ce):.4f}  reg {float(reg):.3e}", flush=True)
