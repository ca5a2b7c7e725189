cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" wb3; do
if [ -n "$v" ]; then export RNDE_LIB=$GRAFT_REPO_ROOT/regneuralde.jl_amd/lib/librnde_$v.so; else unset RNDE_LIB; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_tmp.json 2> gpurun_out/r06_bench_tmp.err
python - "$v" <<'P'
import json, sys
o = json.load(open("gpurun_out/r06_bench_tmp.json"))
r = o["roofline"]
print(repr(sys.argv[1]), {k: round(o[k], 4) for k in ("value", "ms_per_step", "mean_nfe", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms")}, round(r["frac"], 4))
P
done; done
