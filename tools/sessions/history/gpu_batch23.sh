#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/b23_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/b23_pytest.log
run() { python bench.py --workload $1 --steps ${2:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "lr5m_learn: "; run lr5m_learn
echo -n "lr5m_learn NO_KSTAT: "; NSK_DIAG=1 NSK_NO_KSTAT=1 run lr5m_learn
echo -n "boolw4m_learn: "; run boolw4m_learn
echo -n "lr50m_learn: "; run lr50m_learn 5
