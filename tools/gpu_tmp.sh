cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
R=r06 RNDE_COMMIT=767b434 bash tools/gpu_evidence.sh stats pmc pmc4096 sq > gpurun_out/r06/evidence.log 2>&1
{ echo "# clock64 stamps of workgroup 0, diagnostic build (tools/build_diag.sh), MI355X, B = 512, tol 1.4e-8; collected $(date -u '+%Y-%m-%d %H:%M UTC') at 767b434"; for x in 0 1; do echo "## matrix mode $x (RNDE_X3=$x)"; RNDE_X3=$x RNDE_DIAG_BWD=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_bstage.py 2>&1 | grep -v amdgpu.ids | tail -9; done; } > gpurun_out/r06/r06_rev_attempt_stamps.txt
{ echo "# clock64 stamps of workgroup 0 of the one-launch forward solve, diagnostic build; collected $(date -u '+%Y-%m-%d %H:%M UTC') at 767b434"; for x in 0 1; do echo "## matrix mode $x (RNDE_X3=$x)"; RNDE_X3=$x RNDE_DIAG_SOLVE=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_solve.py 2>&1 | grep -v amdgpu.ids | tail -6; done; } > gpurun_out/r06/r06_attempt_stamps.txt
timeout 1500 python bench.py > gpurun_out/r06/r06_bench_line.json 2> gpurun_out/r06/bench.err
tail -c 200 gpurun_out/r06/r06_bench_line.json
cat gpurun_out/r06/r06_rev_attempt_stamps.txt gpurun_out/r06/r06_attempt_stamps.txt | cut -c1-250
