"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per kernel: mean KiB per launch and HBM bytes corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 128-B requests at 64 B on gfx950: double it)."""
import collections, csv, glob, sys
fetch_dir, write_dir = sys.argv[1], sys.argv[2]
def load(d, name):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
F, W = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
import datetime, os
print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- " + os.environ.get("PMC_CMD", "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"))
print("# collected " + datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC") + (" at " + os.environ["RNDE_COMMIT"] if os.environ.get("RNDE_COMMIT") else ""))
print("# corrected bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 : FETCH_SIZE under-reports wide reads 2x on gfx950 (MI355X_MICROARCH.md, HBM)")
print("kernel,dispatches,FETCH_SIZE_mean_KiB,WRITE_SIZE_mean_KiB,hbm_bytes_per_launch_corrected")
rows = []
for k in F:
    f = sum(F[k]) / len(F[k]); w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0])))
    rows.append((len(F[k]) * (2 * f + w), k, len(F[k]), f, w))
for _, k, n, f, w in sorted(rows, reverse=True)[:14]:
    print('"%s",%d,%.1f,%.1f,%d' % (k, n, f, w, int((2 * f + w) * 1024)))
# one launch of rnde_stage_solve_kernel = a whole adaptive solve: attempted steps per launch IN THIS RUN = reversed attempts (one launch each) / solves
ns = sum(len(v) for k, v in F.items() if "rnde_stage_solve_kernel" in k)
nb = sum(len(v) for k, v in F.items() if "rnde_bstage_attempt_kernel" in k)
if ns and nb:
    print('"attempts_per_solve_launch (rnde_bstage_attempt_kernel launches / rnde_stage_solve_kernel launches)",%d,0,0,%.3f' % (ns, nb / ns))
