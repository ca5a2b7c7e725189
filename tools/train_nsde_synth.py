"""End-to-end training of the MNIST Neural SDE on the learnable synthetic set of tools/train_synth.py: vanilla vs `error_est` vs `stiff_est`.

The reference's loop (experiments/mnist_nsde.jl): ClassifierNSDE(Dense(784, 32) -> TrackedNeuralDSDE(Chain(Dense(32, 64, tanh), Dense(64, 32)),
Dense(32, 32), [0, 1], REGULARIZE, solver; reltol = abstol = 1.4e-1) -> Dense(32, 10)) (:70-84), loss = logitcrossentropy + lambda *
mean(sv.saveval) (:88-100), Optimiser(InvDecay(1e-5), ADAM(0.01)) (:86), one trajectory per input in training, `accuracy(...; trajectories = 10)`
for evaluation (:154-155), the NFE probe on a fixed batch (:150-152).  Regularisers as the script selects them from the YAML's `type`
(:45-66): error_est -> SOSRI(), lambda 10, func = EEst * dt; stiff_est (what experiments/configs/mnist_nsde.yml ships) -> AutoSOSRI2(SOSRI2()),
lambda 0.1, func = |eigen_est| / alg_stability_size(SOSRI2()); otherwise SOSRI() without a callback.

    python tools/train_nsde_synth.py [--epochs 10] [--batches 24] [--regs vanilla,error_est,stiff_est] [--out profiles/r05_train_nsde_synth.json]

`func` is passed as a CLOSURE, as the reference's script passes `save_func`: the layer recognises it (node.py::reg_code).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.train_synth import BATCH, NCLS, batches_of, mnist_set, synthetic_set

STAB = 1.0 / 10.6      # `stability_size` of mnist_nsde.jl:54-55


def save_func_error(u, t, integrator):                     # mnist_nsde.jl:48
    return integrator.EEst * integrator.dt


def save_func_stiff(u, t, integrator):                     # mnist_nsde.jl:53-58
    s = abs(integrator.eigen_est)
    return STAB * (0 if (s == 0 or s != s) else s)


SETUPS = {"vanilla": ("SOSRI", 0.0, None), "error_est": ("SOSRI", 10.0, save_func_error), "stiff_est": ("AutoSOSRI2", 0.1, save_func_stiff)}


def accuracy(model, data, trajectories, func):
    hit = n = 0
    with torch.no_grad():
        for xb, yb in data:
            pred = model(xb.reshape(BATCH, -1), trajectories=trajectories, func=func)[0]
            hit += int((pred.argmax(1) == yb.argmax(1)).sum()); n += BATCH
    return 100.0 * hit / n


def run(reg, train, test, epochs, device, seed):
    import regneuralde_jl_amd as rn
    solver, lam, func = SETUPS[reg]
    regularize = func is not None
    g = torch.Generator().manual_seed(seed)
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g), [0.0, 1.0], regularize,
                                solver, save_everystep=False, reltol=1.4e-1, abstol=1.4e-1, save_start=False, max_batch=10 * BATCH, max_attempts=1000, seed=seed)
    model = rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde, rn.Dense(32, 10, "identity", g), device=device)
    opt = rn.FluxADAM(model.trainable(), eta=0.01, gamma=1.0e-5)
    dummy = train[0][0].reshape(BATCH, -1)

    def probe():
        with torch.no_grad():
            _, n1, n2, _ = model(dummy, trajectories=1, func=func)
        return int(n1), int(n2)
    rec = {"regulariser": reg, "solver": solver, "lambda": lam, "epochs": []}
    n1, n2 = probe()
    rec["epochs"].append({"epoch": 0, "nfe1": n1, "nfe2": n2, "test_acc_10_trajectories": accuracy(model, test, 10, func)})
    for epoch in range(1, epochs + 1):
        timing = ce_sum = reg_sum = 0.0
        for xb, yb in train:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                loss, ce, rg, _, _ = rn.fused_nsde_loss_and_grad(model, xb.reshape(BATCH, -1), yb, trajectories=1, lam=lam, regularize=regularize, func=func)
                opt.step()
            except Exception as e:
                rec["failed"] = f"epoch {epoch}: {e}"
                break
            torch.cuda.synchronize()
            timing += time.perf_counter() - t0
            ce_sum += float(ce); reg_sum += float(rg)
        if "failed" in rec:
            break
        n1, n2 = probe()
        e = {"epoch": epoch, "nfe1": n1, "nfe2": n2, "train_acc_1_trajectory": accuracy(model, train[:4], 1, func), "test_acc_10_trajectories": accuracy(model, test, 10, func),
             "train_time_s": timing, "mean_ce": ce_sum / len(train), "mean_reg": reg_sum / len(train), "mean_saveval": (reg_sum / len(train) / lam) if lam else None}
        rec["epochs"].append(e)
        print(f"[{reg:10s}] epoch {epoch:2d}  NFE {n1:4d}/{n2:4d}  test acc (10 traj.) {e['test_acc_10_trajectories']:6.2f}  train time {timing:5.2f} s  ce {e['mean_ce']:.4f}  "
              f"reg {e['mean_reg']:.5f}", flush=True)
    last = rec["epochs"][-1]
    rec["final"] = {k: last.get(k) for k in ("nfe1", "nfe2", "test_acc_10_trajectories", "mean_saveval")}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--test-batches", type=int, default=8)
    ap.add_argument("--regs", default="vanilla,error_est,stiff_est")
    ap.add_argument("--seed", type=int, default=1999)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    device = torch.device("cuda", 0)
    root = os.environ.get("RNDE_MNIST_DIR")
    tr, te, what = mnist_set(root, args.batches * BATCH, args.test_batches * BATCH) if root else synthetic_set(args.batches * BATCH, args.test_batches * BATCH, args.seed)
    train, test = batches_of(*tr, device), batches_of(*te, device)
    out = {"data": what, "batch": BATCH, "train_batches": len(train), "test_batches": len(test), "epochs": args.epochs,
           "loop": "experiments/mnist_nsde.jl:45-66 (regulariser / solver / lambda by `type`), :70-86 (model, optimiser), :88-100 (loss), :150-155 (NFE probe, accuracy with 10 trajectories)",
           "runs": {}}
    for reg in args.regs.split(","):
        out["runs"][reg] = run(reg, train, test, args.epochs, device, args.seed)
    out["summary"] = {k: v["final"] for k, v in out["runs"].items()}
    print(json.dumps(out["summary"], indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
