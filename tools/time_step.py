"""Wall-clock breakdown of one training step (forward solve / head+loss / backward / optimiser)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import regneuralde_jl_amd as rn
from bench import build_model
dev = torch.device("cuda", 0)
B = 512
model = build_model(rn, dev, B)
opt = rn.FluxOptimiser(model.trainable())
g = torch.Generator().manual_seed(1999)
x = torch.rand(B, 1, 28, 28, generator=g).to(dev)
y = torch.eye(10)[torch.randint(0, 10, (B,), generator=g)].to(dev)
def sync(): torch.cuda.synchronize()
acc = {}
def tic(name, t0):
    sync(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
for it in range(8):
    if it == 3: acc.clear()
    sync(); t = time.perf_counter(); t_start = t
    xs = x.reshape(B, -1)
    u, nfe, sv = model.node(xs, model.p2, func="error_est")
    t = tic("1 node forward", t)
    W = model.p3[:7840].view(784, 10); b = model.p3[7840:]
    pred = u @ W + b
    loss = rn.logitcrossentropy(pred, y) + 100.0 * sv.saveval.mean()
    t = tic("2 head + loss", t)
    loss.backward()
    t = tic("3 backward (head + node)", t)
    opt.step()
    t = tic("4 optimiser", t)
    acc["total"] = acc.get("total", 0.0) + (t - t_start)
n = 5
for k in sorted(acc): print(f"{k:28s} {1e3*acc[k]/n:7.3f} ms")
print("nfe", nfe)
