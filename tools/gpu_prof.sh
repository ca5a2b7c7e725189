# rocprofv3 kernel-trace summaries of the three bench workloads -> gpurun_out/${R}prof/*_kernel_stats.csv (copy the ones to be judged into profiles/).
# Run on the GPU box through gpurun:  R=r03 bash tools/gpu_prof.sh
cd $GRAFT_REPO_ROOT
R=${R:-r03}
mkdir -p gpurun_out/${R}prof
export TMPDIR=/tmp
for wl in mnist nsde latent; do
  rm -rf gpurun_out/${R}prof/$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}prof/$wl -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${R}prof/${wl}_bench.log 2>&1
  tail -1 gpurun_out/${R}prof/${wl}_bench.log | cut -c1-300
  head -12 gpurun_out/${R}prof/$wl/*/*_kernel_stats.csv | cut -c1-160
  cp gpurun_out/${R}prof/$wl/*/*_kernel_stats.csv gpurun_out/${R}prof/${wl}_kernel_stats.csv
  find gpurun_out/${R}prof/$wl -name "*kernel_trace.csv" -size +20M -delete
done
