#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q -k "grid or config3 or compact or tally or full_size" > gpurun_out/b22_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/b22_pytest.log
NSK_VARIANTS="new PP1 PP3 OLD new PP1 PP3" bash tools/ab_lib.sh "ising10m ising1m ising40m" 200
