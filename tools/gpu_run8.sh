for x in 0 1; do
echo "== RNDE_X3=$x"
RNDE_X3=$x RNDE_DIAG_SOLVE=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_solve.py 2>&1 | grep "one-launch solve" | tail -1
RNDE_X3=$x timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_x3_$x.json 2> gpurun_out/r06_bench_x3_$x.err
python - <<P
import json
o = json.load(open("gpurun_out/r06_bench_x3_$x.json"))
print("RNDE_X3=$x", {k: o[k] for k in ("value", "ms_per_step", "mean_nfe", "attempts_per_step", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms", "final_loss")})
P
done
