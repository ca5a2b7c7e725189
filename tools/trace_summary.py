"""Timeline summary of a rocprofv3 --kernel-trace csv: per kernel name and grid size, count / mean duration; plus the overlap
of the weight-gradient launches with the reverse sweep.  Usage: trace_summary.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-48:]
    grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    agg[(name, grid)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:14]:
    d = [e - s for s, e in v]
    print(f"{name:48s} grid {str(grid):10s} n {len(v):5d} mean {sum(d)/len(d)/1e3:9.1f} us  min {min(d)/1e3:8.1f} max {max(d)/1e3:8.1f}")
