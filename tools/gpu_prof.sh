cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
tail -1 gpurun_out/prof_bench.log | cut -c1-200
head -14 gpurun_out/prof/*/*_kernel_stats.csv | cut -c1-150
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
