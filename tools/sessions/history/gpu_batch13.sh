#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_config4_gpu.py tests/test_config3_gpu.py -m gpu -q -x > gpurun_out/b13_pytest.log 2>&1
echo "pytest rc $?"; tail -15 gpurun_out/b13_pytest.log
cat gpurun_out/config4_shards_*.json
python tools/lr_tie_calib.py 40000
