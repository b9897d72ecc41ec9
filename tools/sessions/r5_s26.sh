#!/bin/bash
# Round 5, GPU session 26 (runs ON THE GPU BOX): per-group timeline of k_gibbs_ep at the counted-loads library
# (instrumented build libnsk_TIMING.so: s_memtime at a group's start, behind its first pass's rows and at its end).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s26; rm -rf $OUT; mkdir -p $OUT
for N in 5000000 20000000; do
  echo "== $N variables" >> $OUT/timeline.txt
  NSK_LIB=$R/numbskull_amd/variants/libnsk_TIMING.so python tools/timing_ep.py $N >> $OUT/timeline.txt 2> $OUT/err_$N.txt
done
cat $OUT/timeline.txt
