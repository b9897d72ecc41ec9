#!/bin/bash
# round 4, session 30: the GPU suite again after the one configuration assertion of session 29 was brought in
# line with the accumulator-copy rule (the 50M-variable files ran in session 29 and are left out here)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -q --durations=8 --ignore=tests/test_config5_shards_gpu.py --ignore=tests/test_config5_gpu.py > gpurun_out/r4_gpu_tests_s30.log 2>&1
echo "gpu tests rc $? $(tail -1 gpurun_out/r4_gpu_tests_s30.log)"
grep -A10 "slowest" gpurun_out/r4_gpu_tests_s30.log | head -12
