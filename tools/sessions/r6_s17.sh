#!/bin/bash
# Round 6, GPU session 17: the shortened tie tests (200 / 100 epochs) with their statistics printed.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_learning_tie_gpu.py -m gpu -x -q -k "marginals_under or many_weight" -s --durations=6 2>&1 | grep -E "tie:|passed|failed|Error|assert|call" | head -30
