# per-launch durations of the weight-gradient kernels inside the bench, one rocprofv3 kernel trace per library variant:
#   bash tools/gpu_wgrad_times.sh "" sb0 ...      ("" = the default library; NAME = regneuralde.jl_amd/lib/librnde_NAME.so; VAR=VALUE = an environment switch)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
for v in "$@"; do
  rm -rf /tmp/wprof
  unset RNDE_LIB; EXTRA=""
  case "$v" in
    *=*) export "$v"; EXTRA="$v";;
    "") ;;
    *) export RNDE_LIB=$GRAFT_REPO_ROOT/regneuralde.jl_amd/lib/librnde_$v.so;;
  esac
  if [ -n "$PROBE" ]; then timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wprof -- python3 tools/wgrad_probe.py > /tmp/wprof_bench.log 2>&1; else timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wprof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/wprof_bench.log 2>&1; fi
  python3 - "$v" <<'P'
import csv, glob, sys, json
f = glob.glob("/tmp/wprof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
out = {}
for r in rows:
    k = r["Kernel_Name"]
    if "wgrad" not in k or "reduce" in k or "head" in k:
        continue
    key = (k.split("(")[0].replace("void rnde::", ""), int(r["Grid_Size_X"]) // max(1, int(r.get("Workgroup_Size_X") or 448)))
    out.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
js = [l for l in open("/tmp/wprof_bench.log").read().split("\n") if l.startswith("{")]
if js:
    line = json.loads(js[-1])
    print(f"variant '{sys.argv[1]}': ms_per_step {line['ms_per_step']:.3f} rev_rest_ms {line['rev_rest_ms']:.3f} nfe {line['mean_nfe']}")
else:
    print(f"variant '{sys.argv[1]}':", [l for l in open("/tmp/wprof_bench.log").read().split("\n") if l.startswith("attempts")])
for key, v in sorted(out.items()):
    v.sort()
    print(f"   {key[0]:34s} workgroups {key[1]:4d}: n {len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}")
P
  case "$v" in *=*) unset "${v%%=*}";; esac
done
