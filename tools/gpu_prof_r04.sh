# rocprofv3 kernel-trace summary of one bench workload -> gpurun_out/r04prof/<name>_kernel_stats.csv (copy the ones to be judged into profiles/).
#   bash tools/gpu_prof_r04.sh <name> <bench.py arguments...>          (on the GPU box through gpurun; every step is bounded)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=$1; shift
mkdir -p gpurun_out/r04prof
rm -rf gpurun_out/r04prof/$N
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04prof/$N -- python3 bench.py "$@" > gpurun_out/r04prof/${N}_bench.log 2>&1
echo "rocprofv3 rc=$?"
tail -1 gpurun_out/r04prof/${N}_bench.log | cut -c1-400
f=$(find gpurun_out/r04prof/$N -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-170; cp "$f" gpurun_out/r04prof/${N}_kernel_stats.csv; fi
find gpurun_out/r04prof/$N -name "*kernel_trace.csv" -delete
