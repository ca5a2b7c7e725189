for v in 0 15 30 45 60; do
RNDE_WGRAD_SIDE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - "$v" <<'P'
import json, sys
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
print("RNDE_WGRAD_SIDE", sys.argv[1], {k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")})
P
done
