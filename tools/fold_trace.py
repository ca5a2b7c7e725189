"""A window of a rocprofv3 --kernel-trace csv as a timeline, consecutive launches of one kernel folded into one line.
Usage: fold_trace.py <kernel_trace.csv> [from to]   (fractions of the trace's span, default 0.50 0.56)"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:48], r.get("Stream_Id", "?"))
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.50
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.56
t_a = rows[0][0]; span = rows[-1][1] - t_a
sel = [r for r in rows if t_a + lo * span <= r[0] <= t_a + hi * span]
t0 = sel[0][0]; prev_end = t0
i = 0
while i < len(sel):
    j = i
    while j + 1 < len(sel) and sel[j + 1][2] == sel[i][2] and sel[j + 1][3] == sel[i][3] and sel[j + 1][0] - sel[j][1] < 3000:
        j += 1
    run = sel[i:j + 1]
    busy = sum(e - s for s, e, _, _ in run)
    print(f"+{(run[0][0] - t0) / 1e3:9.1f} us  gap {(run[0][0] - prev_end) / 1e3:7.1f}  x{len(run):3d}  span {(run[-1][1] - run[0][0]) / 1e3:8.1f}  busy {busy / 1e3:8.1f}  stream {run[0][3]:>3s}  {run[0][2]}")
    prev_end = max(prev_end, run[-1][1])
    i = j + 1
