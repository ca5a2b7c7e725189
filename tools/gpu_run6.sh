for x in 0 1; do
RNDE_X3=$x timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_x3_$x.json 2> gpurun_out/r06_bench_x3_$x.err
python - <<P
import json
o = json.load(open("gpurun_out/r06_bench_x3_$x.json"))
print("RNDE_X3=$x", {k: o[k] for k in ("value", "ms_per_step", "mean_nfe", "attempts_per_step", "us_per_attempt_fwd", "us_per_attempt_rev", "rev_rest_ms", "final_loss")}, o["roofline"]["us_per_attempt_back_to_back"], o["fixed_weights"])
P
done
