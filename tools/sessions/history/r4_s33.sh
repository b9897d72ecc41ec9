#!/bin/bash
# round 4, session 33: still shorter shape tiles / none at all (the entry-parallel groups take the rest)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for mw in 4 6; do
  echo -n "boolw4m_learn max words $mw: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=$mw one boolw4m_learn
  echo -n "boolw4m max words $mw: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=$mw one boolw4m
done
echo -n "boolw4m_learn no shape tiles: "; NSK_DIAG=1 NSK_NO_SHAPE=1 one boolw4m_learn
echo -n "boolw4m no shape tiles: "; NSK_DIAG=1 NSK_NO_SHAPE=1 one boolw4m
