export RNDE_COMMIT=$(cat .commit 2>/dev/null)
mkdir -p gpurun_out/r06
timeout 200 tools/micro/valu_rate gpurun_out/r06/valu_rate.csv 2>&1 | tee gpurun_out/r06/valu_rate.log
R=r06 bash tools/gpu_coexec.sh > gpurun_out/r06_coexec.log 2>&1
grep "BF16\|^MFMA_ONLY\|^SPLIT \|^LOCKSTEP" gpurun_out/r06/coexec_times.log | head -12
grep "COEXEC" gpurun_out/r06/r06_coexec_micro.csv | tail -6
