mkdir -p gpurun_out/r06b
for x in 0 1; do echo "== RNDE_X3=$x"; RNDE_X3=$x RNDE_DIAG_BWD=1 RNDE_LIB=regneuralde.jl_amd/lib/librnde_diag.so timeout 300 python tools/diag_bstage.py 2>&1 | grep -v amdgpu.ids | tail -14; done > gpurun_out/r06b/rev_stamps.txt 2>&1
cat gpurun_out/r06b/rev_stamps.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_bf16_numerics tools/micro/mfma_bf16_numerics.hip && /tmp/mfma_bf16_numerics gpurun_out/r06b/mfma_bf16_numerics.csv
