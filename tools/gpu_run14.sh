export RNDE_COMMIT=$(cat .commit)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_x3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_prof_bench.log 2>&1
f=$(find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ; collected $(date -u '+%Y-%m-%d %H:%M UTC') at ${RNDE_COMMIT}"; cat "$f"; } > gpurun_out/r06_bench_kernel_stats_mid.csv
head -24 gpurun_out/r06_bench_kernel_stats_mid.csv | cut -c1-230
rm -rf gpurun_out/prof_x3
