timeout 120 tools/micro/mfma_bf16_numerics gpurun_out/r06/mfma_bf16_numerics.csv
for seed in 1999 2000 2001; do for x in 0 1; do
RNDE_X3=$x timeout 600 python tools/train_synth.py --regs vanilla,error_est --seed $seed --out gpurun_out/r06/train_seed${seed}_x3_$x.json 2>&1 | grep -v amdgpu.ids | grep -A12 '^{' | tr -d '\n ' ; echo " <- seed $seed RNDE_X3=$x"
done; done
