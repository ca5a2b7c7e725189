timeout 600 python -m pytest tests/test_gpu_x3.py -q -s 2>&1 | grep -v amdgpu.ids | grep -E "attempts|passed|failed|Error|assert|g_reg" | head -20
for v in "RNDE_X3_WGRAD_OFF=1" "RNDE_X3=1"; do
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - "$v" <<'P'
import json, sys
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
print(sys.argv[1], {k: o[k] for k in ("value", "ms_per_step", "mean_nfe", "attempts_per_step", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms", "final_loss")})
P
done
