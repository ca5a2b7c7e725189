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

# ---- idle gaps: union of kernel intervals over the last 60 % of the trace (the timed steps), largest gaps with their neighbours ----
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows)
t_lo = iv[0][0] + 0.4 * (iv[-1][1] - iv[0][0])
iv = [x for x in iv if x[0] >= t_lo]
busy, gaps, cur_end, last = 0, [], iv[0][0], iv[0][2]
for s, e, nm in iv:
    if s > cur_end:
        gaps.append((s - cur_end, last, nm))
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end, last = e, nm
span = iv[-1][1] - iv[0][0]
print(f"window {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  idle {sum(g[0] for g in gaps)/1e3:.0f} us in {len(gaps)} gaps")
byk = collections.defaultdict(lambda: [0, 0])
for g, a, b2 in gaps:
    byk[(a, b2)][0] += g; byk[(a, b2)][1] += 1
for (a, b2), (g, n) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"   {g/1e3:8.1f} us in {n:4d} gaps  after {a:40s} before {b2}")
