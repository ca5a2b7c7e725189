# Issue-slot accounting of the attempt kernels: SQ counters, one --pmc pass per group (kernel-trace only beside them), per kernel averages
# -> gpurun_out/${R}sq/${R}_pmc_sq_attempt.csv.   R=r03 bash tools/gpu_pmc_sq.sh
cd $GRAFT_REPO_ROOT
R=${R:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out/${R}sq
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/${R}sq/avail.txt
OUT=gpurun_out/${R}sq/${R}_pmc_sq_attempt.csv
echo "# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ; per-dispatch means; collected $(date -u '+%Y-%m-%d %H:%M UTC') at ${RNDE_COMMIT}" > $OUT
echo "kernel,counter,mean_per_dispatch,dispatches" >> $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  d=gpurun_out/${R}sq/$(echo $c | tr ' ' '+')
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $d.log 2>&1
  python3 - "$d" >> $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:48]
    if "stage_attempt" in k or "stage_solve" in k or ("wgrad" in k and "reduce" not in k and "head" not in k):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    for c, x in v.items():
        print('"%s",%s,%.0f,%d' % (k, c, sum(x) / len(x), len(x)))
PY
  find $d -name "*.csv" -size +2M -delete
done
cat $OUT
