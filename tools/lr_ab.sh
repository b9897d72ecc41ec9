#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches']))"; }
for w in lr5m lr5m_learn; do
echo -n "$w default: "; run $w
echo -n "$w SPLIT_GENERAL: "; NSK_DIAG=1 NSK_SPLIT_GENERAL=1 run $w
echo -n "$w NO_BALANCE: "; NSK_DIAG=1 NSK_NO_BALANCE=1 run $w
echo -n "$w GEN_BLOCK=32768: "; NSK_DIAG=1 NSK_GEN_BLOCK=32768 run $w
done
