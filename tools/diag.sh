cd $GRAFT_REPO_ROOT
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_DIAG.so 512 0 2>&1 | grep -v amdgpu.ids | tail -12
python tools/bench_attempt.py regneuralde.jl_amd/lib/librnde_DIAGNODMA.so 512 0 2>&1 | grep -v amdgpu.ids | tail -12
