# rocprofv3 kernel-trace summaries of the three bench workloads (profiles/r02_*): run on the GPU box through gpurun
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02prof
export TMPDIR=/tmp
for wl in mnist nsde latent; do
  rm -rf gpurun_out/r02prof/$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02prof/$wl -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r02prof/${wl}_bench.log 2>&1
  tail -1 gpurun_out/r02prof/${wl}_bench.log | cut -c1-300
  head -16 gpurun_out/r02prof/$wl/*/*_kernel_stats.csv | cut -c1-160
  cp gpurun_out/r02prof/$wl/*/*_kernel_stats.csv gpurun_out/r02prof/${wl}_kernel_stats.csv
  find gpurun_out/r02prof/$wl -name "*kernel_trace.csv" -size +20M -delete
done
