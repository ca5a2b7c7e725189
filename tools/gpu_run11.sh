timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r06_gputests_x3default.log
cat gpurun_out/r06_gputests_x3default.log
