#!/bin/bash
# round 4, session 4: bisect of the 2-rank LR learning mismatch (weight 131) over the diagnostic switches; EP grids
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for sw in NONE NSK_NO_KSTAT NSK_NO_EP NSK_NO_GENERAL NSK_NO_FAST NSK_NO_HEAVY NSK_NO_HUB_EP NSK_NO_PACKED NSK_ONE_ACC NSK_NO_OVERLAP NSK_NO_LEARN_SEG NSK_NO_ZTAB; do
  echo -n "$sw: "
  env LAG=0 NSK_DIAG=1 $sw=1 timeout 300 python tools/debug/lr2.py > gpurun_out/s04_lr2_$sw.log 2>&1
  grep "^sweep" gpurun_out/s04_lr2_$sw.log | grep -v "weight diffs 0" | head -2 | cut -c1-200 | tr '\n' '|'; echo
done
for pcu in 8 10 12 16 24; do
  for w in lr5m lr5m_learn; do
  echo -n "EP_PER_CU $pcu $w: "
  NSK_DIAG=1 NSK_EP_PER_CU=$pcu python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
  done
done
