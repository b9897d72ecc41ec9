#!/bin/bash
# Round 5, GPU session 29 (runs ON THE GPU BOX): k_learn_ep<8> at four waves per SIMD (amdgpu_waves_per_eu(4,4):
# 128 vector registers and 16 bytes of scratch instead of 139 / three waves; libnsk_WPE4.so) against the tree's.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s29; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in lr5m_learn lr50m_learn; do
  case $WL in lr50m*) VS="new WPE4 new WPE4"; S=10; W=3;; *) VS="new WPE4 new WPE4"; S=100; W=10;; esac
  for V in $VS; do
    if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps $S --warmup $W --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_LIB
cat $OUT/bench.txt
