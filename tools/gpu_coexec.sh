# is SQ_VALU_MFMA_COEXEC_CYCLES alive, and do fp32 MFMA / VALU of two waves on one SIMD overlap?   R=r06 bash tools/gpu_coexec.sh
#   wall times of every variant -> gpurun_out/$R/${R}_coexec_micro.csv ; counters per variant (one configuration: ratio 8, K 64) appended to it
cd $GRAFT_REPO_ROOT
R=${R:-r06}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
timeout 120 tools/micro/coexec 256 $O/coexec_times.csv > $O/coexec_times.log 2>&1
cat $O/coexec_times.log
OUT=$O/${R}_coexec_micro.csv
{ echo "# tools/micro/coexec.hip on one MI355X, 256 workgroups of 8 waves (partners w, w + 4 share a SIMD); collected $(date -u '+%Y-%m-%d %H:%M UTC') at ${RNDE_COMMIT}";
  echo "# part 1: times (tools/micro/coexec 256)"; cat $O/coexec_times.csv;
  echo "# part 2: rocprofv3 --kernel-trace --pmc <pair> -- tools/micro/coexec 256 '' 8 64 ; per-dispatch means (nm 4096 MFMAs, nv 32768 FMAs, K 64)";
  echo "kernel,counter,mean_per_dispatch,dispatches"; } > $OUT
for c in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"; do
  d=$O/coexec_$(echo $c | tr ' ' '+')
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- tools/micro/coexec 256 "" 8 64 > $d.log 2>&1
  echo "[$c] rc=$?"
  python3 - "$d" >> $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: sys.exit()
names = {0: "MFMA_ONLY", 1: "VALU_ONLY", 2: "SPLIT", 3: "LOCKSTEP", 4: "STAGGER", 5: "ONEWAVE_MIX", 6: "SPLIT_PRIO_V", 7: "SPLIT_PRIO_M", 8: "SPLIT_NOP", 9: "TWOWAVE_MIX", 10: "ONEWAVE_SEQ", 11: "BF16_MFMA_ONLY", 12: "BF16_SPLIT", 13: "BF16_LOCKSTEP"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "coexec_kernel" not in k: continue
    m = int(k.split("<")[1].split(">")[0])
    agg[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
for m, v in sorted(agg.items()):
    for c, x in v.items():
        print('"coexec_kernel<%d> %s",%s,%.0f,%d' % (m, names[m], c, sum(x) / len(x), len(x)))
PY
  rm -rf $d
done
cat $OUT
