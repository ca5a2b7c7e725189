# HBM traffic of the bench's kernels: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/r02pmc/$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r02pmc/$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02pmc/$c.log 2>&1
  tail -c 300 gpurun_out/r02pmc/$c.log
done
python3 tools/pmc_summary.py gpurun_out/r02pmc/FETCH_SIZE gpurun_out/r02pmc/WRITE_SIZE > gpurun_out/r02pmc/r02_pmc_hbm_traffic.csv
cat gpurun_out/r02pmc/r02_pmc_hbm_traffic.csv
find gpurun_out/r02pmc -name "*.csv" -size +5M -delete
# large per-GPU batch: the same kernels at B = 4096 (roofline at a batch that fills the chip 8 times over)
timeout 900 python3 bench.py --batch 4096 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02pmc/bench_B4096.json 2> gpurun_out/r02pmc/bench_B4096.err
tail -c 2500 gpurun_out/r02pmc/bench_B4096.json
