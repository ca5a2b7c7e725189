timeout 600 python -m pytest tests/test_gpu_x3.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06_gputests_x3default.log
cat gpurun_out/r06_gputests_x3default.log
