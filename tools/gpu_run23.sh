cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
rm -rf gpurun_out/r06b/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06b/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r06b/prof_bench.log 2>&1
cp gpurun_out/r06b/prof/*/*_kernel_stats.csv gpurun_out/r06b/bench_kernel_stats.csv
head -9 gpurun_out/r06b/bench_kernel_stats.csv | cut -c1-230
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r06b/prof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
w = [r for r in rows if "wgrad4x" in r["Kernel_Name"]]
print(len(w), "wgrad4x launches; (grid, duration us) of the last 12:")
for r in w[-12:]:
    print(r["Grid_Size_X"], r.get("Workgroup_Size_X"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:40], r.get("Stream_Id"))
P
find gpurun_out/r06b/prof -name "*kernel_trace.csv" -delete
for v in "RNDE_WGRAD4_CHUNKS=128" "RNDE_WGRAD4_CHUNKS=32" "RNDE_WGRAD_SIDE=0" "RNDE_WGRAD_SIDE=45"; do
echo "== $v"
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - <<'P'
import json
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
r = o["roofline"]
print({k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, round(r["frac"], 4))
P
done
