#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
env LAG=0 NSK_LIB=$R/numbskull_amd/variants/libnsk_LCHECK.so timeout 300 python tools/debug/lr2.py > gpurun_out/s07_lcheck.log 2>&1
grep "MISMATCH\|LCHECKIDS" gpurun_out/s07_lcheck.log | head -20
