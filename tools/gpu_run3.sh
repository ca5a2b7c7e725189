export RNDE_COMMIT=$(cat .commit 2>/dev/null)
mkdir -p gpurun_out/r06
timeout 300 tools/micro/meeting gpurun_out/r06/meeting_times.csv 2>&1 | tee gpurun_out/r06/meeting.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 --share-gpu --global-batch 4096 --no-cpu-baseline --no-extras > gpurun_out/r06_share8.log 2>&1
grep "^{" gpurun_out/r06_share8.log | tail -1 > gpurun_out/r06_share_gpu_8ranks.json
cut -c1-1200 gpurun_out/r06_share_gpu_8ranks.json
grep -i "error\|Traceback" gpurun_out/r06_share8.log | head -5
timeout 1500 python tools/stiff_grad_trained.py --marks 24,72 --pre 96 --sub 50 --out gpurun_out/r06_stiff_grad_trained.json 2>&1 | tail -40
