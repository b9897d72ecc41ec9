#!/bin/bash
# Round 5, GPU session 9 (runs ON THE GPU BOX): where the arrays of the table kernels lie in device memory moved
# their launch time by 3 % (two 4-byte allocations in front of them, r5_s07 / r5_s08).  n tiny allocations in front
# (NSK_ALLOC_PAD=n) against 2 MB-aligned arrays (NSK_ALLOC_ALIGN=1), with the addresses printed.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s09; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/sweep  %.2f us/launch' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['avg_launch_us']))"; }
export NSK_DIAG=1 NSK_VERBOSE=1
for REP in 1 2; do
for CFG in default pad1 pad2 pad3 pad5 align; do
  unset NSK_ALLOC_PAD NSK_ALLOC_ALIGN
  case $CFG in pad*) export NSK_ALLOC_PAD=${CFG#pad};; align) export NSK_ALLOC_ALIGN=1;; esac
  for WL in ising10m ising10m_learn ising40m; do
    [ $REP = 2 ] && [ $WL = ising40m ] && continue
    echo -n "$WL $CFG: " >> $OUT/bench.txt
    python bench.py --workload $WL --steps 200 --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$CFG.err | line >> $OUT/bench.txt
    grep "device arrays" $OUT/${WL}_$CFG.err | tail -1 >> $OUT/bench.txt
  done
done
done
cat $OUT/bench.txt
