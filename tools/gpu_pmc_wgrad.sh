# stall attribution of rnde_wgrad3_kernel: SQ counters, one --pmc pass each (kernel-trace only beside them)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcw
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\(MFMA\|LDS\|WAIT\|BUSY\|CYCLES\)[A-Z_0-9]*" | sort -u > gpurun_out/pmcw/avail.txt
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS"; do
  d=gpurun_out/pmcw/$(echo $c | tr ' ' '+')
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $d.log 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print(d, "no counters"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    if "wgrad3" in k or "stage_attempt" in k:
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
  find $d -name "*.csv" -size +2M -delete
done
