cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_x3.py tests/test_gpu_replay.py tests/test_gpu_coupled.py -x -q -m gpu 2>&1 | tail -3
RNDE_X3=1 RNDE_DIAG_BWD=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_bstage.py 2>&1 | grep -v amdgpu.ids | tail -9
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - <<'P'
import json
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
r = o["roofline"]
print({k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, round(r["frac"], 4))
P
done
