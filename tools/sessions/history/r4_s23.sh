#!/bin/bash
# round 4, session 23: shape classes per id range + an XCD's contiguous eighth of the rest tiles (learning),
# weights kept from pass 1 in shape tiles: the parity suite, then the weighted boolean graph and the LR graph
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch  compile %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s']))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn one part: "; NSK_DIAG=1 NSK_SHAPE_PARTS=1 one boolw4m_learn
echo -n "boolw4m_learn 32 parts: "; NSK_DIAG=1 NSK_SHAPE_PARTS=32 one boolw4m_learn
echo -n "boolw4m: "; one boolw4m
echo -n "boolw4m one part: "; NSK_DIAG=1 NSK_SHAPE_PARTS=1 one boolw4m
echo -n "lr5m_learn: "; one lr5m_learn
echo -n "lr5m: "; one lr5m
echo -n "ising10m_learn: "; one ising10m_learn 100
