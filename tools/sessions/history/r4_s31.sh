#!/bin/bash
# round 4, session 31: words per lane of a shape tile again, now that every rest tile has a wave of its own
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for mw in 12 16 20; do
  echo -n "boolw4m_learn max words $mw: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=$mw one boolw4m_learn
done
echo -n "boolw4m max words 20: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=20 one boolw4m
echo -n "boolw4m_learn 32 parts: "; NSK_DIAG=1 NSK_SHAPE_PARTS=32 one boolw4m_learn
