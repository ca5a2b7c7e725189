# SQ counters of the weight-gradient kernels on the fixed probe workload (tools/wgrad_probe.py), one --pmc pass per pair: gpurun_out/r06b/pmc_wgrad4.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b /tmp/pw
OUT=gpurun_out/r06b/pmc_wgrad4.csv
echo "# rocprofv3 --kernel-trace --pmc <pair> -- python3 tools/wgrad_probe.py --reps 3 ; per-dispatch means of the 256-workgroup launches; collected $(date -u '+%Y-%m-%d %H:%M UTC')" > $OUT
echo "kernel,counter,mean_per_dispatch,dispatches" >> $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  d=/tmp/pw/$(echo $c | tr ' ' '+')
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/wgrad_probe.py --reps 3 > $d.log 2>&1
  python3 - "$d" >> $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:48]
    if "wgrad" in k and "reduce" not in k and "head" not in k and int(r["Grid_Size"]) >= 256 * 448:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    for c, x in v.items():
        print('"%s",%s,%.0f,%d' % (k, c, sum(x) / len(x), len(x)))
PY
done
cat $OUT
