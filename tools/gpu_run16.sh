export RNDE_COMMIT=$(cat .commit)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python bench.py > gpurun_out/r06/r06_bench_line.json 2> gpurun_out/r06/bench.err
python - <<'P'
import json
o = json.load(open("gpurun_out/r06/r06_bench_line.json"))
print({k: o[k] for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, o["roofline"]["frac"], o["roofline"].get("binding", {}).get("frac"), o.get("roofline_B4096", {}).get("frac"), o["cpu_baseline"]["value_min_median_max"])
ow = o.get("other_workloads", {})
print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in ow.items()})
P
R=r06 bash tools/gpu_evidence.sh stats pmc pmc4096 sq > gpurun_out/r06/evidence.log 2>&1
tail -5 gpurun_out/r06/evidence.log
{ echo "# clock64 stamps of workgroup 0 of the one-launch forward solve, diagnostic build (tools/build_diag.sh; RNDE_DIAG_SOLVE=1 RNDE_LIB=.../librnde_diag.so python tools/diag_solve.py), MI355X, B = 512, tol 1.4e-8; collected $(date -u '+%Y-%m-%d %H:%M UTC') at ${RNDE_COMMIT}";
for x in 0 1; do echo "## matrix mode $x (RNDE_X3=$x)"; RNDE_X3=$x RNDE_DIAG_SOLVE=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_solve.py 2>&1 | grep -v amdgpu.ids | tail -5; done; } > gpurun_out/r06/r06_attempt_stamps.txt
cat gpurun_out/r06/r06_attempt_stamps.txt | cut -c1-250
timeout 1500 python tools/train_synth.py --regs vanilla,error_est,stiff_est,stiff_est@0.1 --out gpurun_out/r06/r06_train_synth.json 2>&1 | grep -v amdgpu.ids | tail -14
