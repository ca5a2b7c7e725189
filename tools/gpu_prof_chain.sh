#!/bin/bash
# kernel-trace profile of the config-4 chain engine run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_chain; mkdir -p gpurun_out/prof_chain
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain -- python3 tools/bench_chain.py > gpurun_out/prof_chain/run.log 2>&1
grep -E "attempt|B=" gpurun_out/prof_chain/run.log
head -16 gpurun_out/prof_chain/*/*_kernel_stats.csv | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
