cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python3 tools/bench_attempt.py regneuralde.jl_amd/lib/librnde.so 512 16 > gpurun_out/pmc_sq.log 2>&1
tail -2 gpurun_out/pmc_sq.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "stage_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][17:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v)/len(v) for c, v in d.items()}
    w = m.get("SQ_WAVES", 1)
    print(k, "disp", len(d["SQ_WAVES"]), "waves %.0f" % w, "| per wave: VALU %.0f SALU %.0f VMEM %.0f | wave_cycles(x4) %.0f wait_any %.0f wait_inst %.0f active %.0f" % (
        m["SQ_INSTS_VALU"]/w, m["SQ_INSTS_SALU"]/w, m["SQ_INSTS_VMEM"]/w, 4*m["SQ_WAVE_CYCLES"]/w, 4*m["SQ_WAIT_ANY"]/w, 4*m["SQ_WAIT_INST_ANY"]/w, 4*m["SQ_ACTIVE_INST_ANY"]/w))
PY
find gpurun_out/pmc_sq -name "*.csv" -size +5M -delete
