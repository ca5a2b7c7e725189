cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -4
for v in "" "RNDE_WGRAD4_CHUNKS=128"; do
echo "== $v"
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - <<'P'
import json
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
r = o["roofline"]
print({k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, round(r["frac"], 4))
P
done
rm -rf gpurun_out/r06b/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06b/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06b/prof_bench.log 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r06b/prof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
w = [r for r in rows if "wgrad4x" in r["Kernel_Name"]]
for r in w[-8:]:
    print(r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:40], r.get("Stream_Id"))
P
find gpurun_out/r06b/prof -name "*kernel_trace.csv" -delete
