#!/bin/bash
# Round 5, GPU session 32 (runs ON THE GPU BOX): config #5 at its full size on one GPU, inference and learning
# bit-exact against the oracle, at the round's last library (k_learn_ep_w4).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT0=$R/gpurun_out/r5_s32; rm -rf $OUT0; mkdir -p $OUT0
timeout 560 python -m pytest tests/test_config5_gpu.py -m gpu -x -q > $OUT0/config5.log 2>&1
echo "config5 rc $? $(tail -1 $OUT0/config5.log)"
