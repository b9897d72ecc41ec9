#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x > gpurun_out/b6_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b6_pytest.log
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base: "; run lr5m ""
echo -n "lr5m_learn base: "; run lr5m_learn ""
for v in NOHUB EPNOP1 EPNOP2 NOHUB+EPNOP1+EPNOP2; do
  echo -n "lr5m $v: "; run lr5m $V/libnsk_$v.so
done
NSK_LIB=$V/libnsk_TIMING.so python tools/timing_ep.py 5000000
