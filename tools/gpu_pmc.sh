# HBM traffic of the bench's kernels: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md
# prescribes -> gpurun_out/${R}pmc/${R}_pmc_hbm_traffic.csv.   R=r03 bash tools/gpu_pmc.sh
cd $GRAFT_REPO_ROOT
R=${R:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out/${R}pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${R}pmc/$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${R}pmc/$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${R}pmc/$c.log 2>&1
  tail -c 200 gpurun_out/${R}pmc/$c.log
done
python3 tools/pmc_summary.py gpurun_out/${R}pmc/FETCH_SIZE gpurun_out/${R}pmc/WRITE_SIZE > gpurun_out/${R}pmc/${R}_pmc_hbm_traffic.csv
cat gpurun_out/${R}pmc/${R}_pmc_hbm_traffic.csv
find gpurun_out/${R}pmc -name "*.csv" -size +5M -delete
