"""The stiffness regulariser's gradient WHERE IT MATTERS (reference experiments/mnist_node.jl:70-81: `stiff_est`, lambda 0.1,
`maximum`, AutoTsit5(Tsit5())).  Two questions, one run on the MI355X:

 (a) at TRAINED-LIKE weights (after S optimiser steps of the reference loop on the learnable synthetic set of tools/train_synth.py), how far is the
     device's d(lambda * max_n |eigen_est_n| / 3.5068) / dp from the fp64 restatement?  Both differentiate the SAME discrete program: the fp64 oracle
     replays the device's own (dt, accept) sequence.  Reported per parameter block (W1, b1, W2, b2): |device - fp64|_max / |fp64|_max, the fp32
     oracle's same distance beside it (what ANY fp32 implementation of this term is worth), cosine.
 (b) is the epoch-5 `max_attempts` failure of the lambda = 0.1 run the REGIME or the fp32 GRADIENT?  From the state after `--pre`
     steps the run continues `--sub` steps twice from identical weights and momentum: with the device's gradients, and with the fp64 oracle's
     gradient of the whole loss (CE + lambda * max) substituted into the same optimiser (the forward, hence NFE, stays the device's).

    python tools/stiff_grad_trained.py [--marks 24,72] [--pre 96] [--sub 50] [--out profiles/r06_stiff_grad_trained.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

D, H, NCLS, BATCH = 784, 100, 10, 512
TOL, LAM = 1.4e-8, 0.1
BLOCKS = (("W1", 0, H * (D + 1)), ("b1", H * (D + 1), H * (D + 1) + H), ("W2", H * (D + 1) + H, H * (D + 1) + H + D * (H + 1)),
          ("b2", H * (D + 1) + H + D * (H + 1), H * (D + 1) + H + D * (H + 1) + D))


def per_block(a, ref):
    out = {}
    for name, lo, hi in BLOCKS:
        x, r = a[lo:hi].astype(np.float64), ref[lo:hi].astype(np.float64)
        out[name] = {"rel_max": float(np.abs(x - r).max() / max(np.abs(r).max(), 1e-300)),
                     "rel_l2": float(np.linalg.norm(x - r) / max(np.linalg.norm(r), 1e-300)),
                     "cos": float(x @ r / max(np.linalg.norm(x) * np.linalg.norm(r), 1e-300))}
    x, r = a.astype(np.float64), ref.astype(np.float64)
    out["all"] = {"rel_max": float(np.abs(x - r).max() / np.abs(r).max()), "rel_l2": float(np.linalg.norm(x - r) / np.linalg.norm(r)),
                  "cos": float(x @ r / (np.linalg.norm(x) * np.linalg.norm(r)))}
    return out


def device_steps(L, h, cap):
    steps = (C.c_float * (4 * cap))()
    natt = C.c_int32(0)
    L.rnde_node_steps(h, steps, cap, C.byref(natt))
    return np.array(steps[:4 * natt.value], dtype=np.float32).reshape(-1, 4)


def reg_gradient_check(p2, x, max_attempts):
    """(a): the regulariser term alone (ubar = 0, cotangent lambda on the largest saved value) -- device vs the oracles along the device's steps."""
    from tests.util import Node, Oracle, arch_mnist, make_cfg
    arch = arch_mnist()
    node = Node(make_cfg([D, H, D], ["tanh", "tanh"], BATCH, reltol=TOL, abstol=TOL, regularize=2, max_attempts=max_attempts, col_tile=16))
    got = node.forward(x, p2, 0.0, 1.0, keep_tape=True)
    st = got["steps"]
    dtp, acc = st[:, 1].copy(), st[:, 3].astype(np.int32)
    res = {"attempts": int(len(st)), "nfe": int(got["nfe"]), "saveval_max_device": float(got["saveval"].max())}
    ors = {}
    for name, dt_ in (("f32", np.float32), ("f64", np.float64)):
        o = Oracle(arch, dt_, TOL, TOL, reg_kind=2, max_attempts=max_attempts)
        o.set_replay(dtp.astype(dt_), acc)
        r = o.forward(x.astype(dt_), p2.astype(dt_))
        assert r["rc"] == 0 and len(r["saveval"]) == len(got["saveval"])
        ors[name] = (o, r)
    r64 = ors["f64"][1]
    # (entry 0 is the callback's value at initialisation, eigen_est = 1 -> the constant 1 / 3.5068 [RECALL B.5]: `maximum` picks it -- zero gradient -- while every
    #  step's estimate is below 1; the gradient is measured on the largest STEP value, the entry `maximum` picks once the dynamics have stiffened)
    k = 1 + int(np.argmax(r64["saveval"][1:]))
    res["argmax_step"] = k
    res["init_value"] = float(r64["saveval"][0])
    res["saveval_max_fp64"] = float(r64["saveval"][k])
    res["saveval_rel_err_device"] = float(np.abs(got["saveval"] - r64["saveval"]).max() / np.abs(r64["saveval"]).max())
    res["saveval_rel_err_oracle_f32"] = float(np.abs(ors["f32"][1]["saveval"] - r64["saveval"]).max() / np.abs(r64["saveval"]).max())
    svbar = np.zeros(len(got["saveval"]), np.float32)
    svbar[k] = LAM
    ubar = np.zeros_like(x)
    _, pb_dev, _ = node.backward(ubar, svbar)
    _, pb64, _ = ors["f64"][0].backward(ubar.astype(np.float64), svbar.astype(np.float64))
    _, pb32, _ = ors["f32"][0].backward(ubar, svbar)
    res["device_vs_fp64"] = per_block(pb_dev, pb64)
    res["oracle_f32_vs_fp64"] = per_block(pb32, pb64)
    res["grad_norm_fp64"] = float(np.linalg.norm(pb64))
    node.close()
    return res


def weights_after(marks, batches=24, max_attempts=600):
    """The dynamics' parameters after each of `marks` optimiser steps of the stiff_est loop (lambda 0.1, `maximum`; the first steps of main()'s run) with the batch the
    next step would see: [(steps, p2, x)] -- the states tests/test_gpu_backward.py::test_stiffness_gradient_at_trained_like_weights_is_bounded measures the gradient at."""
    import regneuralde_jl_amd as rn
    from tools.train_synth import batches_of, synthetic_set
    dev = torch.device("cuda", 0)
    tr, _, _ = synthetic_set(batches * BATCH, BATCH, 1999)
    train = batches_of(*tr, dev)
    g = torch.Generator().manual_seed(1999)
    lam0, _, func, agg, solver = rn.REGULARISERS["stiff_est"]
    node = rn.TrackedNeuralODE(rn.MLPDynamics(D, H, generator=g), [0.0, 1.0], True, True, solver, save_everystep=False, reltol=TOL, abstol=TOL, save_start=False,
                               max_batch=BATCH, max_attempts=max_attempts)
    model = rn.ClassifierNODE(node, rn.Dense(D, NCLS, "identity", generator=g), device=dev)
    opt = rn.FluxOptimiser(model.trainable())
    out = []
    for step in range(max(marks) + 1):
        xb, yb = train[step % len(train)]
        if step in marks:
            out.append((step, model.p2.detach().cpu().numpy().copy(), xb.reshape(BATCH, -1).cpu().numpy()))
        if step == max(marks):
            break
        rn.fused_loss_and_grad(model, xb, yb, lam=lam0, regularize=True, sync=True, func=func, agg=agg)
        opt.step()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--marks", default="24,72", help="optimiser steps after which (a) is measured")
    ap.add_argument("--pre", type=int, default=96, help="(b): steps of the lambda = 0.1 run before the two continuations (4 epochs of 24 batches: the failure came in epoch 5)")
    ap.add_argument("--sub", type=int, default=50, help="(b): steps of each continuation; 0 skips (b)")
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--max-attempts", type=int, default=600)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    from tools.train_synth import batches_of, synthetic_set
    from tests.util import Oracle, arch_mnist
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    tr, te, what = synthetic_set(args.batches * BATCH, BATCH, 1999)
    train = batches_of(*tr, dev)
    g = torch.Generator().manual_seed(1999)
    lam0, lam1, func, agg, solver = rn.REGULARISERS["stiff_est"]
    dyn = rn.MLPDynamics(D, H, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, solver, save_everystep=False, reltol=TOL, abstol=TOL, save_start=False, max_batch=BATCH,
                               max_attempts=args.max_attempts)
    model = rn.ClassifierNODE(node, rn.Dense(D, NCLS, "identity", generator=g), device=dev)
    opt = rn.FluxOptimiser(model.trainable())
    marks = sorted(int(m) for m in args.marks.split(",") if m)
    out = {"data": what, "lambda": lam0, "agg": "maximum", "solver": solver, "tol": TOL, "batch": BATCH, "gradient_check": {}, "substitution": None}
    nfes = []

    def device_step(xb, yb):
        loss, ce, rg, nfe = rn.fused_loss_and_grad(model, xb, yb, lam=lam0, regularize=True, sync=True, func=func, agg=agg)
        return float(loss), float(ce), float(rg), int(nfe)

    step = 0
    t0 = time.time()
    while step < max(marks + [args.pre]):
        xb, yb = train[step % len(train)]
        if step in marks:
            p2 = model.p2.detach().cpu().numpy().copy()
            r = reg_gradient_check(p2, xb.reshape(BATCH, -1).cpu().numpy(), args.max_attempts)
            r["weights_norm"] = float(np.linalg.norm(p2))
            out["gradient_check"][str(step)] = r
            print(f"[a] after {step} steps: attempts {r['attempts']} max|eig|/3.5068 {r['saveval_max_fp64']:.4f}  device vs fp64 " +
                  "  ".join(f"{k} {v['rel_max']:.2e}" for k, v in r["device_vs_fp64"].items()) + "  | fp32 oracle vs fp64 " +
                  "  ".join(f"{k} {v['rel_max']:.2e}" for k, v in r["oracle_f32_vs_fp64"].items()), flush=True)
        loss, ce, rg, nfe = device_step(xb, yb)
        opt.step()
        nfes.append(nfe)
        step += 1
        if step % 24 == 0:
            print(f"  step {step}: NFE {nfe} ce {ce:.4f} reg {rg:.4f}  ({time.time() - t0:.0f} s)", flush=True)
    out["pre_nfe_per_step"] = nfes
    if args.sub > 0:
        # ---- (b) two continuations from the same state ----
        state_p = [p.detach().clone() for p in model.trainable()]
        state_v = [v.clone() for v in opt.v]
        state_n = list(opt.n)
        arch = arch_mnist()

        def restore():
            with torch.no_grad():
                for p, s0 in zip(model.trainable(), state_p):
                    p.copy_(s0)
            o = rn.FluxOptimiser(model.trainable())
            o.v = [v.clone() for v in state_v]
            o.n = list(state_n)
            return o

        def continuation(kind):
            o = restore()
            log, failed = [], None
            o64 = Oracle(arch, np.float64, TOL, TOL, reg_kind=2, max_attempts=args.max_attempts) if kind == "fp64" else None
            for i in range(args.sub):
                xb, yb = train[(args.pre + i) % len(train)]
                try:
                    loss, ce, rg, nfe = device_step(xb, yb)
                except Exception as e:
                    failed = f"step {i}: {e}"
                    break
                rec = {"step": i, "nfe": nfe, "ce": ce, "reg": rg}
                if kind == "fp64":      # the fp64 restatement's gradient of the same discrete program (the device's step sequence), CE + lambda * max
                    h = model.node._acquire(xb.reshape(BATCH, -1), True)
                    st = device_steps(L, h.ptr, args.max_attempts)
                    xn, yn = xb.reshape(BATCH, -1).cpu().numpy().astype(np.float64), yb.cpu().numpy().astype(np.float64)
                    p2 = model.p2.detach().cpu().numpy().astype(np.float64)
                    p3 = model.p3.detach().cpu().numpy().astype(np.float64)
                    W, b = p3[:D * NCLS].reshape(D, NCLS), p3[D * NCLS:]
                    o64.set_replay(st[:, 1].astype(np.float64), st[:, 3].astype(np.int32))
                    r = o64.forward(xn, p2)
                    assert r["rc"] == 0
                    logits = r["u"] @ W + b
                    z = logits - logits.max(1, keepdims=True)
                    sm = np.exp(z) / np.exp(z).sum(1, keepdims=True)
                    dl = (sm - yn) / BATCH
                    k = int(np.argmax(r["saveval"]))
                    svbar = np.zeros(len(r["saveval"]))
                    svbar[k] = lam0
                    _, pb, _ = o64.backward(dl @ W.T, svbar)
                    g3 = np.concatenate([(r["u"].T @ dl).reshape(-1), dl.sum(0)])
                    gd = model.p2.grad.detach().cpu().numpy().astype(np.float64)
                    rec["grad_rel_l2_device_vs_fp64"] = float(np.linalg.norm(gd - pb) / np.linalg.norm(pb))
                    rec["grad_cos_device_vs_fp64"] = float(gd @ pb / (np.linalg.norm(gd) * np.linalg.norm(pb)))
                    model.p2.grad = torch.from_numpy(pb.astype(np.float32)).to(dev)
                    model.p3.grad = torch.from_numpy(g3.astype(np.float32)).to(dev)
                o.step()
                log.append(rec)
                if i % 10 == 0 or i == args.sub - 1:
                    print(f"[b:{kind}] step {i}: NFE {nfe} ce {ce:.4f} reg {rg:.4f}" + (f"  |g_dev - g_64| / |g_64| {rec['grad_rel_l2_device_vs_fp64']:.3f} cos {rec['grad_cos_device_vs_fp64']:.4f}"
                                                                                        if kind == "fp64" else ""), flush=True)
            return {"steps": log, "failed": failed, "nfe_first": log[0]["nfe"] if log else None, "nfe_last": log[-1]["nfe"] if log else None,
                    "nfe_max": max(r["nfe"] for r in log) if log else None}
        out["substitution"] = {"from_step": args.pre, "steps": args.sub, "device_gradients": continuation("device"), "fp64_oracle_gradients": continuation("fp64")}
        s = out["substitution"]
        print(json.dumps({k: {kk: v[kk] for kk in ("failed", "nfe_first", "nfe_last", "nfe_max")} for k, v in s.items() if isinstance(v, dict)}, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
