cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_last.log
