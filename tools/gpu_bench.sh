cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -5 | tee gpurun_out/bench_first.log
