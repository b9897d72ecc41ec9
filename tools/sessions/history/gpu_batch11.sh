#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x > gpurun_out/b11_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b11_pytest.log
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d  layoutB %.1f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['layout_bytes_per_update']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base (U2): "; run lr5m ""
echo -n "lr5m U1: "; run lr5m $V/libnsk_U1.so
echo -n "lr5m U3: "; run lr5m $V/libnsk_U3.so
echo -n "lr5m_learn base (U2): "; run lr5m_learn ""
echo -n "lr5m_learn U1: "; run lr5m_learn $V/libnsk_LU1.so
echo -n "lr5m_learn U3: "; run lr5m_learn $V/libnsk_LU3.so
