#!/bin/bash
# Round 5, GPU session 34 (runs ON THE GPU BOX; the round's last GPU minutes): k_gibbs_ep at six waves per SIMD
# (amdgpu_waves_per_eu(6, 6): 80 vector registers, 12-28 bytes of scratch; libnsk_W6.so) against the tree's five.
# Informational: nothing is adopted without a parity run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s34; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in lr5m boolw4m lr50m; do
  case $WL in lr50m*) VS="new W6"; S=5; W=2;; *) VS="new W6 new W6"; S=60; W=10;; esac
  for V in $VS; do
    if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps $S --warmup $W --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
cat $OUT/bench.txt
