#!/bin/bash
# end-of-round evidence: bench line, kernel-trace stats, PMC pass for HBM traffic of the dominant kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final; rm -rf gpurun_out/final/*
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench.log 2>&1
tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench_line.json
cut -c1-300 gpurun_out/final/bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/trace.log 2>&1
head -8 gpurun_out/final/trace/*/*_kernel_stats.csv | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_w.log 2>&1
python3 tools/pmc_summary.py gpurun_out/final/pmc_f gpurun_out/final/pmc_w > gpurun_out/final/pmc_summary.csv 2>&1; head -8 gpurun_out/final/pmc_summary.csv | cut -c1-200
find gpurun_out/final -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/final -name "*counter_collection.csv" -size +30M -delete
