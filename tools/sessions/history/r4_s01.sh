#!/bin/bash
# round 4, session 1: the pairwise peer-to-peer exchange (two processes on one device, 8 handles in one
# process on the 5M LR graph and the 10M grid), captured sequences after a reseed
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/s01_multirank.log 2>&1; echo "multirank rc $?"; tail -3 gpurun_out/s01_multirank.log
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "captured or rccl" > gpurun_out/s01_captured.log 2>&1; echo "captured rc $?"; tail -3 gpurun_out/s01_captured.log
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "lr5m or grid10m" > gpurun_out/s01_shards.log 2>&1; echo "shards rc $?"; tail -5 gpurun_out/s01_shards.log
cat gpurun_out/config5_shards_*.json 2>/dev/null | head -120
