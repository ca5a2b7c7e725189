# step timeline of another bench workload: tools/gpu_timeline_w.sh nsde|latent
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
W=$1
mkdir -p gpurun_out/tl_$W
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$W -o tl -- python3 bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/tl_$W/bench.log 2>&1
f=$(find gpurun_out/tl_$W -name "*kernel_trace.csv" | head -1)
python3 - $f > gpurun_out/tl_$W/timeline.txt <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:48], r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# last 2 % of the trace window, launch by launch
t1 = rows[-1][1]; span = t1 - rows[0][0]
sel = [r for r in rows if r[0] > t1 - 0.03 * span]
t0 = sel[0][0]; prev_end = t0
for s, e, n, st in sel:
    print(f"+{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  stream {st:>3s}  {n}")
    prev_end = max(prev_end, e)
PY
tail -n 120 gpurun_out/tl_$W/timeline.txt
rm -f $f
