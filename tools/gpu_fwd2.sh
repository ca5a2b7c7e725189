cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py -q -x -m gpu 2>&1 | tail -4
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_DIAG.so 512 16 2>&1 | grep -v amdgpu | tail -10
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde.so 512 16 2>&1 | tail -1
