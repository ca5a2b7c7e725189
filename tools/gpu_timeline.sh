cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/tl/bench.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
echo trace $f
python3 tools/step_timeline.py $f 2 > gpurun_out/tl/timeline.txt 2>&1
python3 tools/trace_summary.py $f > gpurun_out/tl/summary.txt 2>&1
head -c 3000 gpurun_out/tl/timeline.txt
rm -f $f   # large
