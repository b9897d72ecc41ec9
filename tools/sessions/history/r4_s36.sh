#!/bin/bash
# round 4, session 36: the rest of the GPU suite at the last library (test_hip_parity ran in session 34, the
# 50M-variable files in session 29), plus the shape-class test as it now stands
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_cabi.py tests/test_config3_gpu.py tests/test_config4_gpu.py tests/test_multirank_gpu.py tests/test_learning_tie_gpu.py -m gpu -q -k "not (test_hip_parity and not shape_classes_per_id_range)" > gpurun_out/r4_gpu_tests_s36.log 2>&1
echo "rc $? $(tail -1 gpurun_out/r4_gpu_tests_s36.log)"
