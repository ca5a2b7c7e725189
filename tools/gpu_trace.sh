#!/bin/bash
# kernel timeline of bench.py (few steps) -> tools/trace_summary.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/trc
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/trc/*/*_kernel_trace.csv
