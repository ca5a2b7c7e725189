timeout 900 python -m pytest tests/test_gpu_x3.py -q -s 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error|assert" | head -20
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12
for v in "RNDE_X3=0" "RNDE_X3_MT=0" "RNDE_X3=1"; do
env $v timeout 300 python bench.py --steps 6 --warmup 2 --batch 4096 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - "$v" <<'P'
import json, sys
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
print("B=4096", sys.argv[1], {k: o[k] for k in ("value", "ms_per_step", "mean_nfe", "attempts_per_step", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")})
P
done
