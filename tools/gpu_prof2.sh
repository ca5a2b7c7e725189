cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -- python3 tools/bench_attempt.py regneuralde.jl_amd/lib/librnde.so 512 16 > gpurun_out/prof2.log 2>&1
tail -2 gpurun_out/prof2.log
head -12 gpurun_out/prof2/*/*_kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof2/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "stage_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-70:]
gaps = [int(tail[i+1]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"]) for i in range(len(tail)-1)]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail]
print("last 70 stage kernels: mean dur %.2f us, mean gap %.2f us" % (sum(durs)/len(durs)/1e3, sum(gaps)/len(gaps)/1e3))
for r, d in list(zip(tail, durs))[:14]: print(r["Kernel_Name"][28:60], d/1e3)
PY
find gpurun_out/prof2 -name "*kernel_trace.csv" -delete
