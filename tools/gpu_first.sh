cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_forward.py -q -s -m gpu 2>&1 | tail -60 > gpurun_out/test_fwd.log
tail -30 gpurun_out/test_fwd.log
