export RNDE_COMMIT=$(cat .commit 2>/dev/null)
R=r06 bash tools/gpu_coexec.sh > gpurun_out/r06_coexec.log 2>&1
cat gpurun_out/r06/coexec_times.log
grep COEXEC gpurun_out/r06/r06_coexec_micro.csv
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 --share-gpu --global-batch 4096 --no-cpu-baseline --no-extras > gpurun_out/r06_share8.log 2>&1
grep "^{" gpurun_out/r06_share8.log | tail -1 > gpurun_out/r06_share_gpu_8ranks.json
cut -c1-1500 gpurun_out/r06_share_gpu_8ranks.json
tail -5 gpurun_out/r06_share8.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_comm.py -x -q 2>&1 | tail -5
