# folded launch timeline of a window of a bench workload's trace: tools/gpu_timeline_w.sh nsde|latent|mnist [from to] (fractions of the trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
W=$1
mkdir -p gpurun_out/tl_$W
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$W -o tl -- python3 bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/tl_$W/bench.log 2>&1
f=$(find gpurun_out/tl_$W -name "*kernel_trace.csv" | head -1)
python3 tools/fold_trace.py $f ${2:-0.50} ${3:-0.56} > gpurun_out/tl_$W/timeline.txt
tail -n 150 gpurun_out/tl_$W/timeline.txt
rm -f $f
