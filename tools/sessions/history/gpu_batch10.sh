#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x > gpurun_out/b10_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b10_pytest.log
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d  layoutB %.1f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['layout_bytes_per_update']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base: "; run lr5m ""
echo -n "lr5m ep_block=512: "; NSK_DIAG=1 NSK_EP_BLOCK=512 run lr5m ""
echo -n "lr5m ep_block=2048: "; NSK_DIAG=1 NSK_EP_BLOCK=2048 run lr5m ""
echo -n "lr5m_learn base: "; run lr5m_learn ""
for v in EPNOP3 NOATOMIC EPNOW EPNOVAL EPNOW+EPNOVAL+EPNOP3; do
  echo -n "lr5m_learn $v: "; run lr5m_learn $V/libnsk_$v.so
done
echo -n "boolw4m: "; run boolw4m ""
echo -n "boolw4m_learn: "; run boolw4m_learn ""
