#!/bin/bash
# Round 6, GPU session 20: shards with wide quads (two shards of a grid, bounds lowered) through the peer-to-peer path.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q -k "two_shards" --durations=5 2>&1 | tail -25
