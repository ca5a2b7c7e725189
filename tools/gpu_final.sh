cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "--- torchrun world=1 (RCCL backend init + all-reduce path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
echo "--- bench"
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json | cut -c1-1200
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
