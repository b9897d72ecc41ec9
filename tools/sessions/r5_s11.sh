#!/bin/bash
# Round 5, GPU session 11 (runs ON THE GPU BOX): k_gibbs_ep on the 50M LR graph (config #5 inference): two rows per
# step (variant EPU2: the 5M sweep of round 3 preferred one, but at 50M the value gathers leave the L2) and more
# workgroups per CU (NSK_EP_PER_CU; the learning launch takes 10 at this size).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s11; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch  gen %.1f s compile %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['generate_s'], d['config']['compile_s']))"; }
export NSK_DIAG=1
for CFG in new EPU2 pcu10 pcu14 EPU2pcu10; do
  unset NSK_LIB NSK_EP_PER_CU
  case $CFG in EPU2*) export NSK_LIB=$R/numbskull_amd/variants/libnsk_EPU2.so;; esac
  case $CFG in *pcu10) export NSK_EP_PER_CU=10;; *pcu14) export NSK_EP_PER_CU=14;; esac
  for WL in lr50m lr5m; do
    [ $WL = lr5m ] && [ $CFG != new ] && [ $CFG != EPU2 ] && continue
    echo -n "$WL $CFG: " >> $OUT/bench.txt
    python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$CFG.err | line >> $OUT/bench.txt
  done
done
cat $OUT/bench.txt
