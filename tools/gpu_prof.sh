cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
tail -1 gpurun_out/prof_bench.log | cut -c1-400
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" | head -20
# keep only summaries small enough to merge back
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
