#!/bin/bash
# round 4, session 24: padded shape classes (null member slots), shape tiles up to 32 words, no uniform
# tiles inside shape classes: parity of the kernel-level tests, then the weighted boolean graph
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_cabi.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch  compile %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s']))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn exact shapes only: "; NSK_DIAG=1 NSK_NO_PAD_SHAPE=1 one boolw4m_learn
echo -n "boolw4m_learn one part: "; NSK_DIAG=1 NSK_SHAPE_PARTS=1 one boolw4m_learn
echo -n "boolw4m_learn 32 parts: "; NSK_DIAG=1 NSK_SHAPE_PARTS=32 one boolw4m_learn
echo -n "boolw4m: "; one boolw4m
echo -n "boolw4m exact shapes only: "; NSK_DIAG=1 NSK_NO_PAD_SHAPE=1 one boolw4m
echo -n "boolw4m one part: "; NSK_DIAG=1 NSK_SHAPE_PARTS=1 one boolw4m
