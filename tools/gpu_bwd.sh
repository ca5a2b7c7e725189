cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -q -s -m gpu > gpurun_out/test_bwd.log 2>&1
grep -E "natt=|passed|failed" gpurun_out/test_bwd.log | grep -v print | tail -40
grep -E "^E  " gpurun_out/test_bwd.log | head -10
