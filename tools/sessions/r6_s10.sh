#!/bin/bash
# Round 6, GPU session 10: per-wave timelines of the wide-quad launch (one and two quads per trip; 10M and 1M grids).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in TIMING TIMING2; do
  echo "=== $v 10M"; NSK_LIB=$R/numbskull_amd/variants/libnsk_$v.so timeout 200 python tools/timing_tabw.py 2500 4000 2>&1 | tail -12
  echo "=== $v 1M"; NSK_LIB=$R/numbskull_amd/variants/libnsk_$v.so timeout 200 python tools/timing_tabw.py 1000 1000 2>&1 | tail -12
done
