"""One training step of a rocprofv3 --kernel-trace csv as a timeline: consecutive launches of the same kernel are folded into one
line (count, span, summed duration, idle time inside the run).  The step is cut at the optimiser kernel.
Usage: step_timeline.py <kernel_trace.csv> [steps-from-the-end, default 2]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:44], r.get("Stream_Id", "?"))
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cuts = [i for i, r in enumerate(rows) if "momentum" in r[2] or "adam" in r[2]]
if len(cuts) < back + 1:
    sys.exit("not enough optimiser launches in the trace")
lo, hi = cuts[-back - 1] + 1, cuts[-back] + 1
step = rows[lo:hi]
t0 = step[0][0]
print(f"step of {len(step)} launches, {(step[-1][1] - t0) / 1e3:.1f} us from first start to last end")
i = 0
while i < len(step):
    j = i
    while j + 1 < len(step) and step[j + 1][2] == step[i][2] and step[j + 1][3] == step[i][3]:
        j += 1
    run = step[i:j + 1]
    dur = sum(e - s for s, e, _, _ in run)
    span = run[-1][1] - run[0][0]
    print(f"  +{(run[0][0] - t0) / 1e3:8.1f} us  {run[0][2]:44s} stream {run[0][3]:>3s}  x{len(run):3d}  span {span / 1e3:8.1f}  busy {dur / 1e3:8.1f}  "
          f"mean {dur / len(run) / 1e3:6.1f}  idle-in-run {(span - dur) / 1e3:6.1f}")
    i = j + 1
