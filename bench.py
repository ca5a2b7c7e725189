#!/usr/bin/env python
"""bench.py -- training-step throughput of the MNIST Neural ODE (BASELINE.json metric).

One "step" = the reference's timed window (experiments/mnist_node.jl:228-234): loss forward (adaptive Tsit5
solve, reltol = abstol = 1.4e-8, error-estimate regulariser), reverse pass through the solver, optimiser
update -- on one batch of 512 synthetic MNIST-shaped images per GPU, already resident in HBM.
N > 1: one process per GPU (torchrun), batch sharded 512 per rank (weak scaling, config "batch 4096 sharded
8xMI355X"), one RCCL all-reduce of the flat gradient (166,418 fp32) per step.

Prints ONE JSON line on rank 0.  Besides the contract's keys:
  value_fixed_weights   the same K steps with the weights restored after every update (the optimiser still runs): NFE does not
                        drift with the number of steps, so this number is comparable across --steps/--warmup choices
  attempts_per_step, us_per_attempt_fwd / _rev, rev_rest_ms, persist_fallback   NFE-independent companions (HIP events)
  roofline      the dominant kernel (the TAPED attempted-step kernel, as a training step runs it) timed live with HIP events
  cpu_baseline  the CPU restatement (oracle/, OpenMP) timed on a bounded sample of the same workload
  other_workloads   N = 1 only: config 4 (latent-ODE dynamics, chain engine) and config 5 (MNIST neural SDE) records
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, NCLS = 784, 100, 10
P_DYN = (D + 1) * H + H + (H + 1) * D + D          # 158,568
ALG_BYTES = lambda B: 34 * 4 * D * B + 6 * 4 * P_DYN   # 34 A + 24 P per attempted step
ALG_FLOPS = lambda B: 6 * 2 * B * ((D + 1) * H + (H + 1) * D)
# matrix mode 1 (csrc/rnde_x3.h): flops the matrix cores actually execute per attempted step -- per 16-column tile and stage 49 row tiles x (4 + 4) k-steps of 32 x 6 terms,
# each a 16 x 16 x 32 instruction (16,384 flop), six stages (the padding of 102 / 112 k-values to 128 and the sixfold split included: it is what the unit does)
X3_MATRIX_FLOPS = lambda B: 6 * ((B + 15) // 16) * 49 * 8 * 6 * 16384
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3
PROFILE_ROUND = "r06"        # profiles/<round>_pmc_hbm_traffic*.csv: the committed --pmc passes `traffic` is read from
ABLATION_ROUND = "r04"       # profiles/<round>_attempt_ablation.csv: ablation of the fp32-input-MFMA attempt kernel (matrix mode 0); the default mode since round 6 runs the matrix cores
MATRIX_BF16_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA


def quiet_gc():
    """Before a timed region: collect once, then move everything alive (torch's ~1 M import-time objects among them) out of the collector's
    sight.  A full collection over those takes ~40 ms -- one of them inside a 20-step timed region of 3.5 ms steps doubled the latent_e2e record in
    four of seven runs of this file (solve_diag showed one 41 ms step whose library calls added up to 3.1 ms).  Measurement hygiene (timeit switches
    the collector off for the same reason); nothing the step does is skipped."""
    import gc
    gc.collect()
    gc.freeze()


def build_model(rn, device, batch, seed=1999):
    import torch
    g = torch.Generator().manual_seed(seed)
    dyn = rn.MLPDynamics(D, H, generator=g)
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], True, True, "Tsit5", save_everystep=False, reltol=1.4e-8,
                               abstol=1.4e-8, save_start=False, max_batch=batch, max_attempts=160)
    post = rn.Dense(D, NCLS, "identity", generator=g)
    model = rn.ClassifierNODE(node, post, device=device)
    return model


def cpu_baseline(batch=512, steps=6, single_thread_batch=128):
    """CPU restatement of the same training step (oracle fp32 + numpy head) on a bounded sample: all host threads on the bench's
    batch, and ONE thread on a quarter of it (SURVEY 8d asks for both).  The oracle's Dense layers are column-blocked
    (oracle/rnde_oracle.c: a weight row is read once per 8 columns, four reverse dot-product chains side by side), which
    leaves every sum in its original order.  `value` is the MEDIAN of the per-step rates; min / median / max are beside it (the GPU
    boxes' 16-thread cgroup quota on 256 visible threads makes single steps vary by +-25 %: quote the range, not one number)."""
    import ctypes
    import numpy as np
    from oracle.oracle import Oracle, arch_mnist, glorot_params
    arch = arch_mnist(D, H)

    def leg(nb, nsteps, threads):
        rng = np.random.default_rng(1999)
        orc = Oracle(arch, np.float32, reltol=1.4e-8, abstol=1.4e-8, reg_kind=1, max_attempts=400)
        if threads:
            orc.lib.orc_set_threads(ctypes.c_int(threads))
        p = glorot_params(arch, rng)
        W3 = rng.uniform(-0.0869, 0.0869, (D, NCLS)).astype(np.float32)
        x = rng.uniform(0, 1, (nb, D)).astype(np.float32)
        y = np.eye(NCLS, dtype=np.float32)[rng.integers(0, NCLS, nb)]

        def step():
            r = orc.forward(x, p)
            logits = r["u"] @ W3
            z = logits - logits.max(1, keepdims=True)
            sm = np.exp(z) / np.exp(z).sum(1, keepdims=True)
            ubar = ((sm - y) / nb) @ W3.T
            svbar = np.full(len(r["saveval"]), 100.0 / len(r["saveval"]), dtype=np.float32)
            orc.backward(ubar.astype(np.float32), svbar)
            return r["nfe"]

        if threads != 1:
            step()
        rates = []
        for _ in range(nsteps):
            t0 = time.perf_counter()
            nfe = step()
            rates.append(nb / (time.perf_counter() - t0))
        rates.sort()
        return rates, int(nfe)

    from oracle.oracle import effective_cores
    cores = effective_cores()                 # affinity capped by the cgroup CPU quota (the GPU boxes: 256 visible, 16 usable)
    r1, nfe1 = leg(single_thread_batch, 1, 1)
    rall, nfe = leg(batch, steps, cores)
    med = rall[len(rall) // 2] if len(rall) % 2 else 0.5 * (rall[len(rall) // 2 - 1] + rall[len(rall) // 2])
    return {"value": med, "unit": "samples/s", "cores": cores, "kind": "port",
            "value_min_median_max": [rall[0], med, rall[-1]],
            "sample": f"{steps} training step(s) timed one by one (value = the median rate), batch {batch} of the same MNIST-NODE workload (fp32 CPU restatement, "
                      f"OpenMP over {cores} threads = the CPUs the cgroup grants of {len(os.sched_getaffinity(0))} visible; NOT the Julia reference, which cannot run here)", "nfe": nfe,
            "single_thread": {"value": r1[0], "unit": "samples/s", "cores": 1,
                              "sample": f"1 training step, batch {single_thread_batch} of the same workload, one thread", "nfe": nfe1}}


def committed_traffic(suffix, wants, rounds):
    """HBM bytes per launch of the first kernel among `wants` found in profiles/<round>_pmc_hbm_traffic<suffix>.csv (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes: tools/pmc_summary.py) -> (bytes, kernel, source) or None."""
    import csv
    for rnd in rounds:
        pf = os.path.join(ROOT, "profiles", f"{rnd}_pmc_hbm_traffic{suffix}.csv")
        if not os.path.exists(pf):
            continue
        rows = [r for r in csv.reader(l for l in open(pf) if not l.startswith("#")) if r]
        head = [l[1:].strip() for l in open(pf) if l.startswith("#")]
        for want in wants:
            for row in rows:
                if want in row[0]:
                    stamp = [l for l in head if l.startswith("collected")]
                    cmd = [l for l in head if "bench.py" in l]
                    return float(row[4]), want, (f"profiles/{rnd}_pmc_hbm_traffic{suffix}.csv (separate --pmc passes of `{cmd[0].split('-- ')[-1] if cmd else 'bench.py'}`, NOT this run"
                                                 + (f"; {stamp[0]}" if stamp else "") + ")")
    return None


def attempt_roofline_at(B, device, steps=3, warmup=2):
    """The roofline unit (one attempted Tsit5 step of the taped forward sweep, timed inside training steps with HIP events) at another
    per-GPU batch: B = 4096 fills the chip eight times over, where the attempt is no latency chain any more."""
    import torch
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    model = build_model(rn, device, B)
    g = torch.Generator().manual_seed(2024)
    x = torch.rand(B, 1, 28, 28, generator=g).to(device)
    y = torch.eye(NCLS)[torch.randint(0, NCLS, (B,), generator=g)].to(device)
    quiet_gc()
    for _ in range(warmup):
        rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
    h = model.node._acquire(x.reshape(B, -1), True)
    L.rnde_node_set_timing(h.ptr, 1)
    fa, rs, atts = [], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        L.rnde_node_timing(h.ptr, C.byref(a), C.byref(b), C.byref(c))
        fa.append(a.value); rs.append(b.value); atts.append(int(L.rnde_node_last_attempts(h.ptr)))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.rnde_node_set_timing(h.ptr, 0)
    us_f, us_r = 1e3 * sum(fa) / max(1, sum(atts)), 1e3 * sum(rs) / max(1, sum(atts))
    t_att = us_f * 1e-6
    out = {"bound": "hbm", "achieved": ALG_BYTES(B) / t_att / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ALG_BYTES(B) / t_att / 1e9 / HBM_PEAK_GBS,
           "traffic": None, "batch": B, "us_per_attempt": us_f, "us_per_attempt_rev": us_r, "attempts_per_step": sum(atts) / len(atts),
           "alg_bytes_per_attempt": ALG_BYTES(B), "mfma_f32_tflops": ALG_FLOPS(B) / t_att / 1e12, "mfma_frac": ALG_FLOPS(B) / t_att / 1e12 / MFMA_F32_PEAK_TF,
           "samples_per_s_fwd_rev_no_update": B * steps / el,
           "kernel": "the same taped attempted-step kernel at a per-GPU batch of %d (loss forward + reverse, no optimiser update, fixed weights)" % B}
    if B == 4096:      # the committed PMC passes of `bench.py --batch 4096` (two column tiles per workgroup, one launch per attempted step)
        tr = committed_traffic("_B4096", ["rnde_stage_attempt_mt_kernel", "rnde_stage_attempt_kernel"], (PROFILE_ROUND,))
        if tr is not None:
            out["traffic"], out["traffic_kernel"], out["traffic_source"] = tr[0], tr[1] + " (1 launch = 1 attempted step)", tr[2]
    del model
    torch.cuda.empty_cache()
    return out


def global_batch_anchor(G, device, steps, warmup):
    """The FULL training step (loss forward, reverse pass, InvDecay / Momentum update) at a batch of G on ONE GPU: the N = 1 point of the
    strong-scaling curve `python bench.py --gpus N --global-batch G` measures (BASELINE.json: batch 4096 sharded 8 x MI355X).  Run with the SAME
    --steps / --warmup as the N-rank line and in the same two legs (weights restored after every update first, then training), so that both ends of
    the ratio have seen the same number of updates: NFE drifts while the weights train, and a ratio across different NFE measures the drift."""
    import torch
    import regneuralde_jl_amd as rn
    model = build_model(rn, device, G)
    opt = rn.FluxOptimiser(model.trainable())
    g = torch.Generator().manual_seed(1999)
    x = torch.rand(G, 1, 28, 28, generator=g).to(device)
    y = torch.eye(NCLS)[torch.randint(0, NCLS, (G,), generator=g)].to(device)
    nfes = []
    saved = [p.detach().clone() for p in model.trainable()]

    def step(restore):
        loss, ce, reg, nfe = rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=False)
        opt.step()
        if restore:
            with torch.no_grad():
                for p, s0 in zip(model.trainable(), saved):
                    p.copy_(s0)
        nfes.append(nfe)

    def leg(restore):
        quiet_gc()
        for _ in range(warmup):
            step(restore)
        nfes.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(restore)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"value": G * steps / el, "ms_per_step": 1e3 * el / steps, "mean_nfe": sum(nfes) / len(nfes)}

    fixed = leg(True)
    opt = rn.FluxOptimiser(model.trainable())
    train = leg(False)
    out = {"value": train["value"], "unit": "samples/s", "ms_per_step": train["ms_per_step"], "mean_nfe": train["mean_nfe"],
           "value_fixed_weights": fixed["value"], "fixed_weights": fixed, "steps": steps, "warmup": warmup,
           "global_batch": G, "n_gpus": 1, "what": f"full training step (with optimiser update) at batch {G} on one GPU, same --steps / --warmup and the same two legs as "
                                                    f"the headline: divide `value` (or `value_fixed_weights`: equal NFE by construction) of `bench.py --gpus N "
                                                    f"--global-batch {G} --steps {steps} --warmup {warmup}` by the same field here for the strong-scaling factor"}
    del model, opt
    torch.cuda.empty_cache()
    return out


def bench_latent(args, B=512, throughput_record=True):
    """SURVEY.md 8d config 4 ("latent ODE dynamics only, for the kernel benchmark"): gen_dynamics of experiments/latent_ode.jl:113-124
    (tanh + 8 x Dense(20<->50, tanh), P = 8,280), B = 512, 49 saveat points on [0, 1], Tsit5 at 1.4e-8; forward + reverse of the layer call."""
    import torch
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    T = 49
    g = torch.Generator().manual_seed(1999)
    dyn = rn.LatentGenDynamics(generator=g)
    grid = [i / (T - 1) for i in range(T)]
    node = rn.TrackedNeuralODE(dyn, [0.0, 1.0], False, True, "Tsit5", saveat=grid, reltol=1.4e-8, abstol=1.4e-8, max_batch=B, max_attempts=256)
    z0 = torch.randn(B, 20, generator=g).to(device).requires_grad_(True)
    p = node.p.to(device).clone().requires_grad_(True)
    w = torch.randn(B, T, 20, generator=g).to(device)

    def step_autograd():
        z0.grad = None; p.grad = None
        res, nfe, sv = node(z0, p)
        ((res * w).sum() / B + 10.0 * sv.saveval.mean()).backward()
        return nfe
    # the same forward + reverse as two calls through the C ABI (what a host that owns its own tape would make): the scalar loss
    # sum(res * w) / B + 10 * mean(saveval) has the cotangents w / B and 10 / n, no tape library, no device <-> host tensor traffic
    L = _lib.lib()
    node._func = "error_est"
    hd = node._acquire(z0.detach(), True)
    zc, pc = z0.detach().contiguous(), p.detach().contiguous()
    res = torch.empty(B, T, 20, dtype=torch.float32, device=device)
    ubar = (w / B).contiguous()
    zbar, pbar = torch.empty_like(zc), torch.empty_like(pc)
    sa = (C.c_float * T)(*grid)
    sv_host = (C.c_float * (node.max_attempts + 1))()
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def step_abi():
        nfe, nsv = C.c_int64(0), C.c_int32(0)
        _lib.check(hd.ptr, L.rnde_node_forward_saveat(hd.ptr, zc.data_ptr(), pc.data_ptr(), B, 0.0, 1.0, sa, T, res.data_ptr(), C.byref(nfe), sv_host,
                                                      C.byref(nsv), 1, stream))
        n = nsv.value
        svb = (C.c_float * n)(*([10.0 / n] * n))
        _lib.check(hd.ptr, L.rnde_node_backward_async(hd.ptr, ubar.data_ptr(), svb, zbar.data_ptr(), pbar.data_ptr(), None, stream))
        return int(nfe.value)
    step = step_autograd if args.autograd else step_abi
    quiet_gc()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nfes = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the roofline unit: one attempted step as the timed steps run it -- TAPED, inside a solve (HIP events around the attempt launches of
    # three more steps, rnde_node_set_timing); the untaped forced attempt of the micro-benchmark is reported beside it
    L.rnde_node_set_timing(hd.ptr, 1)
    fa, atts = [], []
    for _ in range(3):
        step_abi()
        torch.cuda.synchronize()
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        L.rnde_node_timing(hd.ptr, C.byref(a), C.byref(b), C.byref(c))
        fa.append(a.value); atts.append(int(L.rnde_node_last_attempts(hd.ptr)))
    L.rnde_node_set_timing(hd.ptr, 0)
    us_in_solve = 1e3 * sum(fa) / max(1, sum(atts))
    h = node._acquire(z0.detach(), False)
    us = C.c_float(0)
    _lib.check(h.ptr, L.rnde_bench_attempt(h.ptr, z0.detach().contiguous().data_ptr(), p.detach().data_ptr(), B, 200, C.byref(us), None))
    flops = 6 * 2 * B * 8280
    one_launch = int(L.rnde_node_one_launch_solves(hd.ptr)) > 0
    del node, hd, h
    torch.cuda.empty_cache()
    big = None
    if throughput_record and B == 512 and not args.autograd:
        # THROUGHPUT MODE (round 4): the same one-launch solve and one-launch reverse sweep with 256 workgroups (B = 4096: every CU holds one, they meet
        # through agent-scope entries once per attempt) -- what the engine delivers when more columns exist than the 32 tiles of the latency case
        class _A: pass
        a2 = _A(); a2.steps, a2.warmup, a2.autograd = max(5, args.steps // 2), max(2, args.warmup // 2), False
        try:
            r = bench_latent(a2, B=4096, throughput_record=False)
            big = {k: r[k] for k in ("value", "unit", "ms_per_step", "mean_nfe", "steps", "warmup")}
            big.update({"batch": 4096, "workgroups": 256, "us_per_attempt": r["roofline"]["us_per_attempt"], "mfma_frac": r["roofline"]["frac"],
                        "one_launch_solve": r["one_launch_solve"], "speedup_over_B512": r["value"] / (B * args.steps / el)})
        except Exception as e:
            big = {"error": repr(e)}
    return {"metric": "forward+reverse samples/sec, latent-ODE dynamics (config 4)", "value": B * args.steps / el, "unit": "samples/s",
            "one_launch_solve": one_launch, "throughput_B4096": big,
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mean_nfe": sum(nfes) / len(nfes),
            "config": {"workload": f"latent ODE gen_dynamics D=20, 8 Dense layers 20<->50 tanh, B={B}, 49 saveat points, Tsit5 1.4e-8 (chain engine); step = layer forward (taped) + its reverse, "
                                   + ("through torch.autograd" if args.autograd else "two C-ABI calls (rnde_node_forward_saveat, rnde_node_backward_async)")},
            "roofline": {"bound": "mfma", "achieved": flops / (us_in_solve * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flops / (us_in_solve * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                         "kernel": "rnde_chainmw_kernel<2,MW_SOLVE,0,1> (multi-wave chain engine, weights register stationary): the WHOLE adaptive solve = 1 launch; "
                                   "unit = one attempted Tsit5 step inside it, taped, timed over the solves of training steps (the reverse sweep is one launch too, "
                                   "rnde_bchainmw_kernel<2,0,1,1>); us_per_attempt_forced_untaped = one attempt as its own launch (rnde_chainmw_kernel<2,0,0,1>); "
                                   f"latency bound ({(B + 15) // 16} workgroups of 4 waves on the chip)",
                         "us_per_attempt": us_in_solve, "us_per_attempt_forced_untaped": us.value}}


def bench_latent_e2e(args):
    """Config 4 END TO END: the whole latent-ODE training step of experiments/latent_ode.jl:339-349 on synthetic PhysioNet-shaped
    series (75 x 49 x 512: 37 data rows, 37 mask rows at 30 % observed, one time-difference row) -- recognition GRU (49 steps), rec_to_gen, sampling,
    the layer call with 49 save times (Tsit5, 1.4e-8), gen_to_data, masked likelihood + KL + lambda_r mean(EEst dt), the reverse of all of it, and
    Optimiser(InvDecay(1e-5), AdaMax(0.01)) -- every piece through the C ABI (rn.fused_latent_loss_and_grad + FluxAdaMax -> rnde_adamax_step)."""
    import torch
    import regneuralde_jl_amd as rn
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, T = 512, 49
    g = torch.Generator().manual_seed(1999)
    grid = torch.linspace(0, 1, T)
    model = rn.build_latent_ode(saveat=grid, regularize=True, generator=g, device=device, max_batch=B, max_attempts=256)
    data = torch.randn(B, T, 37, generator=g).to(device)
    mask = (torch.rand(B, T, 37, generator=g) < 0.3).float().to(device)
    mask[:, 0, 0] = 1.0
    t_row = torch.full((B, T, 1), 1.0 / (T - 1)).to(device); t_row[:, -1] = 0.0
    opt = rn.FluxAdaMax(model.trainable())
    nfes = []
    model._trace = []      # host time of each library call, per step (solve_diag.slowest_step_calls_ms)

    def step():
        total, nll, kl, reg, nfe = rn.fused_latent_loss_and_grad(model, data, mask, t_row, lam_r=1.0e3, lam_k=1.0, generator=None)
        opt.step()
        nfes.append(nfe)
        return total

    quiet_gc()
    for _ in range(args.warmup):
        step()
    nfes.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        last = step()
        step_ms.append(1e3 * (time.perf_counter() - ts))      # (host time of the step's calls: the forward's host wait is in it, the asynchronous reverse is not)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    from regneuralde_jl_amd import _lib as _l
    hn = next(iter(model.node._handles.values()))[0]
    solve_diag = {"one_launch_solves": int(_l.lib().rnde_node_one_launch_solves(hn.ptr)), "fallback_count": int(_l.lib().rnde_node_fallback_count(hn.ptr)),
                  "forward_calls": args.warmup + args.steps, "host_ms_per_step_min_median_max": [round(min(step_ms), 3), round(sorted(step_ms)[len(step_ms) // 2], 3), round(max(step_ms), 3)],
                  "host_ms_per_step": [round(v, 2) for v in step_ms], "nfe_per_step": list(nfes),
                  "slowest_step_calls_ms [encode, layer forward, decode + loss, layer reverse (async), encode reverse]": model._trace[args.warmup + step_ms.index(max(step_ms))],
                  "median_step_calls_ms": model._trace[args.warmup + step_ms.index(sorted(step_ms)[len(step_ms) // 2])]}
    # the pieces around the solve on their own (HIP events on the launch stream): encode and its reverse, decode + loss
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    import ctypes as Cc
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    hl = model._latent_handle
    x_ = torch.cat([data, mask, t_row], dim=2).contiguous()
    p1, p2, p3, p4 = (p.detach() for p in model.trainable())
    eps = torch.randn(B, 20, device=device)
    z0, mu0, lv = (torch.empty(B, 20, device=device) for _ in range(3))
    res, resb = torch.randn(B, T, 20, device=device), torch.empty(B, T, 20, device=device)
    loss2, p4b, p1b, p2b = torch.empty(2, device=device), torch.empty_like(p4), torch.empty_like(p1), torch.empty_like(p2)
    st = Cc.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    reps = 10
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps):
        L.rnde_latent_encode(hl.ptr, x_.data_ptr(), p1.data_ptr(), p2.data_ptr(), eps.data_ptr(), B, T, z0.data_ptr(), mu0.data_ptr(), lv.data_ptr(), st)
        L.rnde_latent_decode_loss(hl.ptr, res.data_ptr(), p4.data_ptr(), x_.data_ptr(), B, T, loss2.data_ptr(), resb.data_ptr(), p4b.data_ptr(), st)
        L.rnde_latent_encode_backward(hl.ptr, z0.data_ptr(), 1.0, p1.data_ptr(), p2.data_ptr(), x_.data_ptr(), p1b.data_ptr(), p2b.data_ptr(), st)
    ev[1].record(); torch.cuda.synchronize()
    around_ms = ev[0].elapsed_time(ev[1]) / reps
    return {"metric": "latent ODE, FULL model training step (config 4 end to end)", "value": B * args.steps / el, "unit": "samples/s",
            "ms_per_step": 1e3 * el / args.steps, "mean_nfe": sum(nfes) / len(nfes), "steps": args.steps, "warmup": args.warmup, "final_loss": float(last),
            "ms_per_step_median_host": sorted(step_ms)[len(step_ms) // 2],      # (a one-off ~40 ms step -- a buffer of the solve growing while the step count creeps up -- shows in the mean of a short run, not here)
            "solve_diag": solve_diag, "ms_around_the_solve": around_ms, "step_over_solve_part": (1e3 * el / args.steps) / max(1e-9, 1e3 * el / args.steps - around_ms),
            "what_is_around": "rnde_latent_encode (49-step GRU, rec_to_gen, sampling) + rnde_latent_decode_loss (gen_to_data, likelihood, reverse) + "
                              "rnde_latent_encode_backward (reverse GRU, 8 weight-gradient GEMMs), HIP events, without the layer call between them",
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LatentTimeSeriesModel: LatentGRU(37, 40, 50), rec_to_gen 100-50-40, gen_dynamics 8 x Dense(20 <-> 50, tanh) over 49 save times, "
                                   "gen_to_data 20-37; batch 512, Tsit5 reltol = abstol = 1.4e-8, error_est regulariser, InvDecay + AdaMax; the weights train",
                       "global_batch": B, "parallelism": "single"}}


def bench_nsde_stiff(args):
    """Config 5 as the reference SHIPS it (experiments/configs/mnist_nsde.yml:6 `type: stiff_est`): AutoSOSRI2(SOSRI2()), the callback records
    |eigen_est| / 10.6, lambda = 0.1 on its mean (mnist_nsde.jl:51-61)."""
    return bench_nsde(args, reg_type="stiff_est")


def bench_nsde(args, reg_type=None):
    """BASELINE config 5: MNIST neural SDE (experiments/mnist_nsde.jl:70-100), D = 32, drift 32 -> 64 -> 32, diffusion 32 -> 32, SOSRI at
    reltol = abstol = 0.14, B = 512, trajectories = 1 (training); step = ClassifierNSDE loss forward (Dense(784,32) -> one-launch adaptive
    solve -> Dense(32,10), logitcrossentropy + 10 * mean(EEst*dt)) + reverse + ADAM(0.01) update."""
    import torch
    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = 512
    reg_type = reg_type or getattr(args, "nsde_type", None) or "error_est"
    stiff = reg_type == "stiff_est"
    solver, lam = ("AutoSOSRI2", 0.1) if stiff else ("SOSRI", 10.0)       # mnist_nsde.jl:45-61: (SOSRI(), lambda 10) / (AutoSOSRI2(SOSRI2()), lambda 0.1)
    g = torch.Generator().manual_seed(1999)
    nsde = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g), [0.0, 1.0], True,
                                solver, save_everystep=False, reltol=0.14, abstol=0.14, save_start=False, max_batch=B, max_attempts=256, seed=1999)
    model = rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde, rn.Dense(32, 10, "identity", g), device=device)
    opt = torch.optim.Adam(model.trainable(), lr=0.01) if args.autograd else rn.FluxADAM(model.trainable(), eta=0.01, gamma=1.0e-5)   # Optimiser(InvDecay(1.0e-5), ADAM(0.01)), mnist_nsde.jl
    x = torch.rand(B, 784, generator=g).to(device)
    y = torch.eye(NCLS)[torch.randint(0, NCLS, (B,), generator=g)].to(device)
    L = _lib.lib()
    stats = {"att": [], "acc": [], "solve_ms": [], "rev_ms": []}
    nsde_step = [0]
    last_reg = [0.0]

    def step(record):
        if args.autograd:
            opt.zero_grad(set_to_none=True)
            loss, ce, reg, nfe1, nfe2 = rn.nsde_loss_function(x, y, model, trajectories=1, lam=lam, func=reg_type)
            loss.backward()
            # Optimiser(InvDecay(1e-5), ADAM(0.01)): InvDecay scales the gradient by 1 / (1 + gamma n) in front of ADAM (the C-ABI leg's rn.FluxADAM does the same)
            nsde_step[0] += 1
            with torch.no_grad():
                for p_ in model.trainable():
                    if p_.grad is not None:
                        p_.grad.mul_(1.0 / (1.0 + 1.0e-5 * nsde_step[0]))
        else:   # the same loss and gradients without a tape library in the loop (nsde.fused_nsde_loss_and_grad)
            loss, ce, reg, nfe1, nfe2 = rn.fused_nsde_loss_and_grad(model, x, y, trajectories=1, lam=lam, func=reg_type)
        opt.step()
        if record:
            last_reg[0] = float(reg)
            h = next(iter(nsde._handles.values()))[0]
            a, b, na, nc = C.c_float(0), C.c_float(0), C.c_int32(0), C.c_int32(0)
            L.rnde_nsde_timing(h.ptr, C.byref(a), C.byref(b), C.byref(na), C.byref(nc))
            stats["att"].append(na.value); stats["acc"].append(nc.value); stats["solve_ms"].append(a.value); stats["rev_ms"].append(b.value)
        return nfe1, nfe2
    quiet_gc()
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nf = [step(False) for _ in range(args.steps)]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for _ in range(3):
        step(True)
    att = sum(stats["att"]) / len(stats["att"])
    acc = sum(stats["acc"]) / len(stats["acc"])
    us_att = 1e3 * sum(stats["solve_ms"]) / sum(stats["att"])
    flops = 8 * B * (32 * 64 + 64 + 64 * 32 + 32 + 32 * 32 + 32)      # 4 drift + 4 diffusion evaluations per attempted step
    # THROUGHPUT MODE (round 4): the reference's evaluation call, accuracy(...; trajectories = 10) (mnist_nsde.jl:154-155): the batch of 512 images
    # expanded to 5,120 columns in ONE solve with ONE error norm (supervised_classification.jl:87-98) -- 320 workgroups of the four-wave kernel
    ev = None
    try:
        nsde2 = rn.TrackedNeuralDSDE(rn.Chain(rn.Dense(32, 64, "tanh", g), rn.Dense(64, 32, "identity", g)), rn.Dense(32, 32, "identity", g), [0.0, 1.0], True,
                                     solver, save_everystep=False, reltol=0.14, abstol=0.14, save_start=False, max_batch=10 * B, max_attempts=256, seed=1999)
        model2 = rn.ClassifierNSDE(rn.Dense(784, 32, "identity", g), nsde2, rn.Dense(32, 10, "identity", g), device=device)
        with torch.no_grad():
            for k_, v_ in zip(model2.trainable(), model.trainable()):
                k_.copy_(v_)                       # the weights the timed steps above trained
            reps = max(5, args.steps // 2)
            for _ in range(3):
                model2(x, trajectories=10, func=reg_type)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nfe_e = [model2(x, trajectories=10, func=reg_type)[1] for _ in range(reps)]
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
        h2 = next(iter(nsde2._handles.values()))[0]
        a, b, na, nc = C.c_float(0), C.c_float(0), C.c_int32(0), C.c_int32(0)
        L.rnde_nsde_timing(h2.ptr, C.byref(a), C.byref(b), C.byref(na), C.byref(nc))
        ev = {"what": "ClassifierNSDE evaluation call, trajectories = 10: Dense(784,32) -> ONE adaptive solve over 5,120 columns (320 workgroups, one error norm) -> "
                      "Dense(32,10) -> mean over trajectories; forward only, as accuracy() runs it",
              "images_per_s": B * reps / el2, "columns_per_s": 10 * B * reps / el2, "ms_per_call": 1e3 * el2 / reps, "mean_nfe1": sum(nfe_e) / len(nfe_e),
              "attempts_last_call": na.value, "solve_ms_last_call": a.value, "us_per_attempt": 1e3 * a.value / max(1, na.value), "columns": 10 * B, "workgroups": 10 * B // 16,
              "columns_per_s_of_the_B512_training_forward": B / (1e-3 * sum(stats["solve_ms"]) / 3)}
    except Exception as e:
        ev = {"error": repr(e)}
    return {"metric": "training-step samples/sec + NFE, MNIST Neural SDE bs=512 (config 5)", "evaluation_5120_columns": ev, "value": B * args.steps / el, "unit": "samples/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (library Philox noise)",
            "mean_nfe1": sum(a for a, _ in nf) / len(nf), "mean_nfe2": sum(b for _, b in nf) / len(nf),
            "attempts_per_step": att, "accepted_per_step": acc, "solve_ms": sum(stats["solve_ms"]) / 3, "rev_sweep_ms": sum(stats["rev_ms"]) / 3,
            "config": {"workload": f"MNIST NSDE regularized ({reg_type}, lambda {lam:g}), {'AutoSOSRI2(SOSRI2())' if stiff else 'SOSRI'} reltol=abstol=0.14, B=512, trajectories=1, "
                                   "diagonal noise; step = ClassifierNSDE loss fwd + reverse + InvDecay/ADAM update"},
            "mean_saveval_last_step": last_reg[0] / lam if lam else None,
            "roofline": {"bound": "mfma", "achieved": flops / (us_att * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flops / (us_att * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                         "kernel": "rnde_sde_solve_mw_kernel: the WHOLE adaptive solve = 1 launch, one workgroup of four waves per 16-column tile; "
                                   "unit = one attempted SRI step inside it (8 small network evaluations + the cross-workgroup norm); "
                                   "latency bound (32 workgroups on the chip)",
                         "us_per_attempt": us_att}}


def spawn_multi_gpu(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh torchrun child (nothing here has touched the GPU
    yet, and the child is a child process, not an exec) and pass its JSON line through."""
    import socket
    with socket.socket() as so:                 # a port that is free right now (a fixed or pid-derived one can sit in TIME_WAIT)
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fixed-weights leg and the config 4 / config 5 records")
    ap.add_argument("--col-tile", type=int, default=0)
    ap.add_argument("--autograd", action="store_true", help="head + loss through torch.autograd instead of the fused C-ABI head")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and run the gradient all-reduce even at world size 1: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--coupled", action="store_true", help="SURVEY 8e mode 2: ONE step-size controller for all ranks (rnde_node_set_coupling: an all-reduce of the error-norm "
                    "partials after every attempted step; reproduces the single-device run at the global batch; default: independent controllers)")
    ap.add_argument("--share-gpu", action="store_true", help="test rig: every rank on device 0, torch.distributed on gloo, the gradient collective = the library's one-shot "
                    "kernel over peer-mapped windows (no RCCL: it refuses two ranks on one GPU); the N > 1 code path on a 1-GPU box, not a throughput claim")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: the global batch is fixed (e.g. 4096) and split evenly over the ranks "
                    "(per-rank batch = G / world, \"scaling\": \"strong\"); default 0 = weak scaling at --batch per rank")
    ap.add_argument("--nsde-type", default="error_est", choices=["error_est", "stiff_est"], help="--workload nsde: the regulariser (experiments/configs/mnist_nsde.yml `type`; the reference ships stiff_est)")
    ap.add_argument("--workload", default="mnist", choices=["mnist", "latent", "latent_e2e", "nsde"], help="mnist = BASELINE.json's metric (default); latent = config 4 (dynamics only); "
                    "latent_e2e = config 4 with the whole model around the solve; nsde = config 5")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_multi_gpu(args))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus} (or run `python bench.py --gpus N` directly)")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the integration path has no CPU fallback")
    if args.workload == "latent":
        return print(json.dumps(bench_latent(args)), flush=True)
    if args.workload == "nsde":
        return print(json.dumps(bench_nsde(args)), flush=True)
    if args.workload == "latent_e2e":
        return print(json.dumps(bench_latent_e2e(args)), flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    ddev = torch.device("cpu") if args.share_gpu else device      # where the bench's own bookkeeping tensors live

    import regneuralde_jl_amd as rn
    from regneuralde_jl_amd import _lib
    L = _lib.lib()
    B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the {world} ranks")
        B = args.global_batch // world
    rig_note = None
    if args.share_gpu:
        rig_note = "share-gpu: ALL ranks on one device over gloo + peer windows -- exercises the N > 1 code path, NOT a throughput figure"
        if world * 7 * ((B + 15) // 16) > 256 and "RNDE_PERSIST" not in os.environ:
            # the one-launch solve / attempt kernels need all 7 x ceil(B / 16) workgroups of a rank resident at once; `world` ranks of them do not fit the 256 CUs of
            # the ONE device they share here, and a rank whose workgroups wait for CUs another rank's spinning kernel holds gives up (bounded) -- with the
            # asynchronous reverse pass that is an error one step later.  On the rig (never on a rank that owns its GPU) those ranks run launch by launch.
            os.environ["RNDE_PERSIST"] = "0"
            rig_note += f"; persistent kernels OFF on this rig ({world} ranks x {7 * ((B + 15) // 16)} workgroups > 256 CUs of the one shared device: 7 launches per attempted step)"
    model = build_model(rn, device, B)
    model.node.col_tile = args.col_tile
    opt = rn.FluxOptimiser(model.trainable())
    g = torch.Generator().manual_seed(1999 + rank)
    x = torch.rand(B, 1, 28, 28, generator=g).to(device)                  # uniform [0,1) images, mnist_node.jl:206
    y = torch.eye(NCLS)[torch.randint(0, NCLS, (B,), generator=g)].to(device)
    fg = rn.FlatGrads(model.trainable()) if use_dist else None
    reducer = rn.GradientAllReducer(model.trainable(), flat=fg, collective="peers" if args.share_gpu else None) if use_dist else None   # the library's communicator (rnde_comm_*: RCCL unless RNDE_COLLECTIVE / RNDE_ONESHOT say otherwise)
    if args.share_gpu and reducer is not None and world > 4:
        # RIG ONLY: more than four ranks' queues on one device -- a rank's spinning one-shot kernel can wait 20 s for a peer whose own all-reduce sits behind
        # compute kernels the device does not get to (seen at world 8: the first all-reduce gave up).  On the rig every rank drains its stream and meets
        # the others on the host before the collective is enqueued, so the eight all-reduce kernels are the only work in flight (what tools/oneshot_worker.py
        # does between its all-reduces); the step runs as separate library calls (RNDE_ONE_CALL=0).  Ranks that own their GPU never take this path.
        os.environ["RNDE_ONE_CALL"] = "0"
        inner_allreduce = reducer.allreduce_range_

        def rig_allreduce(lo, hi, mean=False):
            torch.cuda.synchronize()
            dist.barrier()
            return inner_allreduce(lo, hi, mean)
        reducer.allreduce_range_ = rig_allreduce
        rig_note += "; host barrier in front of every gradient all-reduce (world > 4 on one device)"
    if args.coupled:
        if reducer is None or reducer.comm is None:
            raise SystemExit("--coupled needs the library communicator: run with --gpus N > 1 (or --force-dist)")
        model.node.set_coupling(reducer.comm, world * B)
    nfes = []
    saved = [p.detach().clone() for p in model.trainable()]

    def train_step(restore=False):
        if args.autograd:
            loss, ce, reg, nfe = rn.loss_function(x, y, model, lam=1.0e2)
            loss.backward()
            loss = float(loss.detach())
            if use_dist:
                reducer.allreduce_(mean=False)
        else:   # loss stays on the device; with a reducer the two gradient all-reduces (RCCL over xGMI, 166,418 fp32 in all) are queued inside
            loss, ce, reg, nfe = rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=False, flat=fg, reducer=reducer)
        opt.step(grad_scale=reducer.grad_scale if use_dist else 1.0)
        if restore:                                                       # fixed-weights leg: the update ran, its effect is undone
            with torch.no_grad():
                for p, s0 in zip(model.trainable(), saved):
                    p.copy_(s0)
        nfes.append(nfe)
        return loss

    def timed(restore):
        nfes.clear()
        quiet_gc()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = train_step(restore)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([el], device=ddev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
            nf = torch.tensor([sum(nfes) / len(nfes)], device=ddev, dtype=torch.float64)
            dist.all_reduce(nf)
            mean_nfe = float(nf.item()) / world
        else:
            mean_nfe = sum(nfes) / len(nfes)
        return el, mean_nfe, float(last)

    # fixed-weights leg FIRST (from the initial weights: stationary by construction), then the real training leg
    fixed = None
    if not args.no_extras:
        for _ in range(args.warmup):
            train_step(True)
        el_f, nfe_f, _ = timed(True)
        fixed = {"value": world * B * args.steps / el_f, "ms_per_step": 1e3 * el_f / args.steps, "mean_nfe": nfe_f}
        opt = rn.FluxOptimiser(model.trainable())                         # fresh optimiser state for the training leg
    quiet_gc()
    for _ in range(args.warmup):
        train_step()
    elapsed, mean_nfe, last_loss = timed(False)

    # ---- N > 1 diagnostics (every rank takes part): per-rank NFE and persistent-kernel fallbacks, and the all-reduce on its own (HIP events) ----
    dist_diag = None
    if use_dist:
        hh = model.node._acquire(x.reshape(B, -1), True)
        mine = torch.tensor([float(sum(nfes) / max(1, len(nfes))), float(L.rnde_node_fallback_count(hh.ptr)), float(L.rnde_node_launches_per_attempt(hh.ptr))],
                            device=ddev, dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        def time_allreduce(red, reps=20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                red.allreduce_range_(0, fg.flat.numel())
            dist.barrier(); torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                red.allreduce_range_(0, fg.flat.numel())
            e1.record(); torch.cuda.synchronize()
            fg.flat.zero_()
            return 1e3 * e0.elapsed_time(e1) / reps

        ar_us, ar_both = None, None
        if reducer is not None and reducer.comm is not None:
            ar_us = time_allreduce(reducer)
            ar_both = {L.rnde_comm_path(reducer.comm).decode(): ar_us}
        nf = [float(v[0]) for v in allv]
        dist_diag = {"nfe_per_rank": nf, "nfe_min": min(nf), "nfe_mean": sum(nf) / len(nf), "nfe_max": max(nf),
                     "persist_fallback_count_per_rank": [int(v[1]) for v in allv], "launches_per_attempt_per_rank": [int(v[2]) for v in allv],
                     "allreduce_us": ar_us, "allreduce_us_by_path": ar_both, "allreduce_floats": int(fg.flat.numel()) if fg is not None else None,
                     "collective_library": L.rnde_comm_library().decode() if reducer is not None and reducer.comm is not None and reducer.collective == "rccl" else None,
                     # (RNDE_ONESHOT=1 in the environment: the one-shot kernel over peer-mapped windows instead of ncclAllReduce)
                     "collective_path": L.rnde_comm_path(reducer.comm).decode() if reducer is not None and reducer.comm is not None else None,
                     "collective_fallback": getattr(reducer, "fallback_reason", None) if reducer is not None else None}

    out = None
    if args.coupled:
        model.node.set_coupling(None, 0)     # the legs below run on rank 0 alone: a shared controller would wait there for the other ranks
    if rank == 0:
        h = model.node._acquire(x.reshape(B, -1), True)
        # --- NFE-independent companions: HIP events around the attempted steps of the forward and the reverse sweep, 3 steps ---
        L.rnde_node_set_timing(h.ptr, 1)
        fa, rs, rr, atts = [], [], [], []
        for _ in range(3):
            rn.fused_loss_and_grad(model, x, y, lam=1.0e2, sync=True)
            a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
            L.rnde_node_timing(h.ptr, C.byref(a), C.byref(b), C.byref(c))
            fa.append(a.value); rs.append(b.value); rr.append(c.value); atts.append(int(L.rnde_node_last_attempts(h.ptr)))
        L.rnde_node_set_timing(h.ptr, 0)
        # --- roofline leg: the TAPED attempted-step kernel alone (what a training step runs), HIP events on the launch stream ---
        us, us_untaped = C.c_float(0), C.c_float(0)
        xs = x.reshape(B, -1).contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        stage_engine = args.col_tile in (0, 16)
        if stage_engine:    # back to back, every attempt into ANOTHER tape record (32 in turn), as in a solve: the tape does not stay on the chip
            _lib.check(h.ptr, L.rnde_bench_attempt_cold_tape(h.ptr, xs.data_ptr(), model.p2.data_ptr(), B, 200, 32, C.byref(us), stream))
        else:
            _lib.check(h.ptr, L.rnde_bench_attempt_taped(h.ptr, xs.data_ptr(), model.p2.data_ptr(), B, 200, C.byref(us), stream))
        _lib.check(h.ptr, L.rnde_bench_attempt(h.ptr, xs.data_ptr(), model.p2.data_ptr(), B, 200, C.byref(us_untaped), stream))
        nl = int(L.rnde_node_launches_per_attempt(h.ptr))
        one_launch = int(L.rnde_node_one_launch_solves(h.ptr)) > 0      # the forward solve ran as ONE launch (rnde_stage_solve_kernel)
        # The roofline unit is timed where the contract asks: INSIDE training steps, HIP events on the launch stream around the forward
        # sweep (3 steps above), divided by the number of attempted steps (= us_per_attempt_fwd; rocprofv3's per-kernel average of this
        # command is a little lower because it averages the ~4.5 us early-exit launches in).  The back-to-back micro-benchmark is reported
        # beside it: an attempt is ~2 us faster there, because its prologue finds controller state and starting record warm.
        us_in_step = 1e3 * sum(fa) / max(1, sum(atts))      # sweep time / REAL attempts (the ~4.5 us early-exit launch behind a solve is charged to them)
        t_att = us_in_step * 1e-6
        roof = {"bound": "hbm", "achieved": ALG_BYTES(B) / t_att / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ALG_BYTES(B) / t_att / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "kernel": (("rnde_stage_solve_kernel (taped, timed inside training steps): ONE launch = the whole adaptive solve; the roofline unit stays one attempted Tsit5 "
                            "step = launch duration / attempts of that solve (units_per_launch)" if one_launch else
                            "rnde_stage_attempt_kernel (taped, timed inside training steps): one attempted Tsit5 step = 1 launch (7 stages, in-kernel slab hand-off)" if nl == 1 else
                            "rnde_stage_kernel (taped): one attempted Tsit5 step = 7 launches (START, 5 x STAGE, LAST)") if stage_engine
                           else "rnde_step_kernel: one attempted Tsit5 step = 1 launch"),
                "launches_per_unit": (1.0 / max(1.0, sum(atts) / len(atts))) if one_launch else nl,
                "units_per_launch": (sum(atts) / len(atts)) if one_launch else 1.0 / nl,
                "us_per_launch": (1e3 * sum(fa) / len(fa)) if one_launch else us_in_step / nl,
                "us_per_attempt": us_in_step, "us_per_attempt_back_to_back": us.value, "us_per_attempt_back_to_back_untaped": us_untaped.value,
                "alg_bytes_per_attempt": ALG_BYTES(B), "alg_bytes_per_launch": ALG_BYTES(B) * (sum(atts) / len(atts)) if one_launch else ALG_BYTES(B) / nl,
                "mfma_f32_tflops": ALG_FLOPS(B) / t_att / 1e12,
                "mfma_frac": ALG_FLOPS(B) / t_att / 1e12 / MFMA_F32_PEAK_TF,
                # Which roof actually binds.  HBM does not (0.39 x the algorithmic bytes reach it: the state lives in registers / L2).  Round 6 settled the unit question
                # (profiles/r06_coexec_micro.csv): on gfx950 the fp32-input MFMA executes on the VECTOR ALUs (64 FLOP/clk/SIMD, no overlap with the vector instructions
                # of any wave on its SIMD: two waves take the SUM of their times; the same pair with a bf16 MFMA takes the MAXIMUM and SQ_VALU_MFMA_COEXEC_CYCLES counts
                # it) -- so matrix mode 0 was bound by that shared unit.  Matrix mode 1 (default: csrc/rnde_x3.h) forms every fp32 product from six bf16 matrix-core
                # products: `binding` is then the stage's own critical path -- vector issue (two tanh per stage, the splitting) + the hand-off latency + the matrix
                # pipe, which the lock-step of a workgroup's waves keeps from overlapping fully.  `frac` above stays the north star's unit.
                "matrix_mode": int(L.rnde_node_matrix_mode(h.ptr)),
                "binding": ({"bound": "stage critical path: vector issue + hand-off latency + matrix-core pipe (bf16x3, six v_mfma_f32_16x16x32_bf16 per 32 k-values)",
                             "achieved": X3_MATRIX_FLOPS(B) / t_att / 1e12, "peak": MATRIX_BF16_PEAK_TF, "unit": "TFLOP/s (bf16 matrix-core flops actually issued)",
                             "frac": X3_MATRIX_FLOPS(B) / t_att / 1e12 / MATRIX_BF16_PEAK_TF,
                             "fp32_equivalent_tflops": ALG_FLOPS(B) / t_att / 1e12,
                             "evidence": "profiles/r06_coexec_micro.csv (fp32 MFMA = vector-ALU instruction; bf16 MFMA co-executes), profiles/r06_attempt_stamps.txt (cycle stamps "
                                         "of both matrix modes), profiles/r06_pmc_sq_attempt.csv"}
                            if int(L.rnde_node_matrix_mode(h.ptr)) == 1 else
                            {"bound": "fp32 vector ALUs (the fp32-input MFMA and the vector instructions share them: profiles/r06_coexec_micro.csv)", "achieved": ALG_FLOPS(B) / t_att / 1e12,
                             "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ALG_FLOPS(B) / t_att / 1e12 / MFMA_F32_PEAK_TF,
                             "evidence": "profiles/r05_pmc_sq_attempt.csv (fp32 MFMA busy 32 %), profiles/r04_attempt_ablation.csv (MFMAs alone: 13.8 us of the 23.3)"})}
        # HBM traffic per launch of that kernel: PMC counters cannot be collected from inside the bench; the committed separate
        # passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command, corrected as MI355X_MICROARCH.md prescribes) are
        # reported when present, with their source.  `traffic` is per LAUNCH like `achieved`; the one-launch solve's launch holds
        # `units_per_launch` attempted steps of the PROFILED run (its step count differs from this run's: the per-attempt figure is the comparable one).
        try:
            if B == 512:      # the committed passes are of the B = 512 bench
                wants = (["rnde_stage_solve_kernel"] if one_launch else []) + (["rnde_stage_attempt_kernel"] if (stage_engine and nl == 1) else []) + ["rnde_stage_kernel<1, 1>"]
                tr = committed_traffic("", wants, (PROFILE_ROUND, "r03") if not one_launch else (PROFILE_ROUND,))
                if tr is not None:
                    roof["traffic"], roof["traffic_source"] = tr[0] * (7 if tr[1].startswith("rnde_stage_kernel") else 1), tr[2]
                    if tr[1] == "rnde_stage_solve_kernel":
                        pa = committed_traffic("", ["attempts_per_solve_launch"], (PROFILE_ROUND,))
                        if pa is not None:      # per LAUNCH of THIS run: the profiled launch held pa[0] attempted steps, this run's holds units_per_launch
                            roof["traffic_per_attempt"] = tr[0] / pa[0]
                            roof["traffic_profiled_launch"] = {"bytes": tr[0], "attempts": pa[0]}
                            roof["traffic"] = roof["traffic_per_attempt"] * roof["units_per_launch"]
        except Exception:
            pass
        # the measured floor of this kernel's decomposition: its MFMAs alone with operands in registers (profiles/r0N_attempt_ablation.csv)
        try:
            import csv
            for row in csv.DictReader(l for l in open(os.path.join(ROOT, "profiles", ABLATION_ROUND + "_attempt_ablation.csv")) if not l.startswith("#")):
                if row["variant"] == "mfmaonly" and int(row["B"]) == B:
                    roof["ceiling_us"] = float(row["us_taped"])
                    roof["ceiling_frac"] = ALG_BYTES(B) / (roof["ceiling_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                    roof["ceiling_source"] = "profiles/" + ABLATION_ROUND + "_attempt_ablation.csv: the fp32-input-MFMA attempt kernel's MFMAs alone (matrix mode 0: operands in registers, no polls / tanh / LDS / tape), same launch geometry, back to back -- NOT a bound of matrix mode 1, whose products run on the matrix cores"
        except Exception:
            pass
        out = {"metric": "training-step samples/sec + mean NFE, MNIST Neural ODE bs=512",
               "value": world * B * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "mean_nfe": mean_nfe, "final_loss": last_loss,
               # what this line can NOT say (no Julia runtime, no MNIST files, no network in the build or on the GPU box): the device is checked against the builder's
               # CPU restatement of the published algorithms, never against a run of the reference itself
               "parity_vs_reference": "unpinned (device == oracle/ restatement at the reference tolerance; no Julia-produced vector exists: SURVEY 8c, DESIGN.md 2)",
               "accuracy_vs_reference": "unmeasured (no MNIST, no Julia; synthetic learnable set only: DESIGN.md 7)",
               "value_fixed_weights": None if fixed is None else fixed["value"],
               "fixed_weights": fixed,
               "attempts_per_step": sum(atts) / len(atts),
               "us_per_attempt_fwd": 1e3 * sum(fa) / max(1, sum(atts)),
               "us_per_attempt_rev": 1e3 * sum(rs) / max(1, sum(atts)),
               "rev_rest_ms": sum(rr) / len(rr),
               "persist_fallback": bool(stage_engine and nl != 1), "persist_fallback_count": int(L.rnde_node_fallback_count(h.ptr)),
               "controller": "coupled (one controller for all ranks, SURVEY 8e mode 2)" if args.coupled else "independent per rank (SURVEY 8e mode 1)",
               "collective": (None if not use_dist else "rnde_comm_allreduce (librnde.so; path: see dist.collective_path): ONE sum-all-reduce of the flat gradient buffer [p2-bar | p3-bar] "
                              "behind the reverse pass, on the compute stream; 1/world folded into the optimiser launch"),
               "rig": rig_note,
               "config": {"workload": f"MNIST NODE regularized (error_est), Tsit5 reltol=abstol=1.4e-8, batch {B} per GPU, "
                                      "1xMI355X per rank; step = loss fwd + reverse pass through the solver + "
                                      "InvDecay/Momentum update; the weights train during the timed steps (mean_nfe drifts with "
                                      "--steps: value_fixed_weights is the stationary companion)"
                                      + (f"; STRONG scaling: global batch {args.global_batch} split over {world} rank(s)" if args.global_batch else ""), "global_batch": world * B,
                          "parallelism": f"dp{world}" if world > 1 else "single"},
               "roofline": roof}
        if dist_diag is not None:
            out["dist"] = dist_diag
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_extras and not use_dist and B == 512:
            try:
                out["roofline_B4096"] = attempt_roofline_at(4096, device)
            except Exception as e:
                out["roofline_B4096"] = {"error": repr(e)}
        if world == 1 and not args.no_extras and not use_dist and B == 512:
            try:      # the N = 1 anchor of the strong-scaling curve (north star: batch 4096 over 1 / 2 / 4 / 8 GPUs): the FULL training step at B = 4096
                others_anchor = global_batch_anchor(4096, device, args.steps, args.warmup)
            except Exception as e:
                others_anchor = {"error": repr(e)}
        else:
            others_anchor = None
        if world == 1 and not args.no_extras and not use_dist:
            # (the secondary records run the headline's --steps / --warmup: with 2 warm-up steps a one-off tape reallocation -- ~40 ms, the step count creeps up
            #  while the weights train -- landed inside a 10-step timed region in two of four runs and doubled the latent_e2e figure; `solve_diag` shows the spread)
            sub = argparse.Namespace(steps=args.steps, warmup=args.warmup, autograd=args.autograd)
            others = {}
            for name, fn in (("latent_config4", bench_latent), ("latent_e2e", bench_latent_e2e), ("nsde_config5", bench_nsde), ("nsde_config5_stiff_est", bench_nsde_stiff)):
                try:
                    others[name] = fn(sub)
                except Exception as e:      # a secondary record must never cost the headline line
                    others[name] = {"error": repr(e)}
            if others_anchor is not None:
                others["global4096"] = others_anchor
            out["other_workloads"] = others
    # ---- the OTHER collective path in the same run (so that one multi-GPU lease decides the default): the one-shot kernel over peer-mapped windows
    # (hipIpc) next to RCCL.  It has never met more than one GPU, so it runs LAST, behind everything the line reports, under a watchdog: if the probe has
    # not come back in time rank 0 prints the line as it stood before the probe and every rank leaves with status 3 (a wedged collective is not a success).  After every phase all ranks agree (MIN all-reduce)
    # before any of them goes on, so a failure on one rank cannot strand the others in a barrier.
    if (use_dist and dist_diag is not None and reducer is not None and reducer.comm is not None and reducer.collective == "rccl"
            and not os.environ.get("RNDE_ONESHOT") and os.environ.get("RNDE_BENCH_BOTH_COLLECTIVES", "1") != "0"):
        import copy
        import threading
        by_path = dist_diag["allreduce_us_by_path"]
        # the line as it stands BEFORE the probe, serialised now: the watchdog thread never walks a dictionary the main thread may be writing
        snap = None
        if rank == 0:
            snap = copy.deepcopy(out)
            snap["dist"]["allreduce_us_by_path"]["one_shot_error"] = "watchdog: the probe of the one-shot collective did not return in 90 s; the process left with status 3"
            snap = json.dumps(snap)

        def give_up():      # a wedged collective is a FAILURE of the run: the measured line is still printed (it was complete before the probe), the status says what happened
            try:
                if rank == 0:
                    C.CDLL(None).fflush(None)
                    print(snap, flush=True)
            finally:
                os._exit(3)
        dog = threading.Timer(90.0, give_up)
        dog.daemon = True
        dog.start()

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ddev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1
        alt, why = None, None
        try:
            alt = rn.GradientAllReducer(model.trainable(), flat=fg, collective="peers")
        except Exception as e:
            why = repr(e)
        if not agree(alt is not None and alt.comm is not None):
            by_path["one_shot_error"] = why or "another rank could not map the peer windows"
        else:
            n_fl = fg.flat.numel()
            ramp = torch.arange(n_fl, device=fg.flat.device, dtype=torch.float32).remainder_(97.0)
            exact, why = False, None
            try:      # the same sum on every rank, bit for bit (small integers: exact in fp32)
                fg.flat.copy_(ramp + float(rank))
                alt.allreduce_range_(0, n_fl)
                torch.cuda.synchronize()
                exact = bool(torch.equal(fg.flat, ramp * world + world * (world - 1) / 2.0))
            except Exception as e:
                why = repr(e)
            if not agree(why is None):
                by_path["one_shot_error"] = why or "the one-shot all-reduce failed on another rank"
            else:
                us, why = None, None
                try:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for _ in range(3):
                        alt.allreduce_range_(0, n_fl)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        alt.allreduce_range_(0, n_fl)
                    e1.record(); torch.cuda.synchronize()
                    us = 1e3 * e0.elapsed_time(e1) / 20
                except Exception as e:
                    why = repr(e)
                if agree(why is None):
                    by_path[L.rnde_comm_path(alt.comm).decode()] = us
                    by_path["one_shot_sum_exact"] = exact
                else:
                    by_path["one_shot_error"] = why or "the timed one-shot all-reduces failed on another rank"
            fg.flat.zero_()
        dog.cancel()
        del alt
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        C.CDLL(None).fflush(None)          # RCCL's version banner sits in the C stdio buffer: push it out BEFORE the JSON line
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
