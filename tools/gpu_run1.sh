export RNDE_COMMIT=$(cat .commit 2>/dev/null)
R=r06 bash tools/gpu_coexec.sh > gpurun_out/r06_coexec.log 2>&1
tail -30 gpurun_out/r06_coexec.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_1.log
cat gpurun_out/r06_gputests_1.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 --share-gpu --global-batch 4096 --no-cpu-baseline --no-extras > gpurun_out/r06_share8.log 2>&1
grep "^{" gpurun_out/r06_share8.log | tail -1 > gpurun_out/r06_share_gpu_8ranks.json
cut -c1-1500 gpurun_out/r06_share_gpu_8ranks.json
