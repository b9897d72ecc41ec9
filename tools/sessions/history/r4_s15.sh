#!/bin/bash
# round 4, session 15: weights with one factor updated in place; lag only for LDS-accumulating handles
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_multirank_gpu.py tests/test_learning_tie_gpu.py tests/test_config3_gpu.py -m gpu -x -q > gpurun_out/s15_parity.log 2>&1; echo "parity rc $?"; tail -4 gpurun_out/s15_parity.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
one() { python bench.py --workload $1 --steps ${2:-50} --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch  ok %s clipped %s' % (d['value'], d['roofline']['avg_launch_us'], d['parity']['ok'], d.get('learn_clipped')))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn (NSK_NO_DIRECT): "; NSK_DIAG=1 NSK_NO_DIRECT=1 one boolw4m_learn
echo -n "lr5m_learn: "; one lr5m_learn
echo -n "ising10m_learn: "; one ising10m_learn 100
