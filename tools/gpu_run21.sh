timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - <<'P'
import json
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
r = o["roofline"]
print({k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, round(r["us_per_attempt_back_to_back"], 3), round(r["us_per_attempt_back_to_back_untaped"], 3), round(r["frac"], 4))
P
done
