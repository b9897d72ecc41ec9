#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base: "; run lr5m ""
for v in EPNOVAL EPNOP1 EPNOP2 NODRAW EPNOP1+EPNOP2+NODRAW; do echo -n "lr5m $v: "; run lr5m $V/libnsk_$v.so; done
echo -n "lr5m_learn base: "; run lr5m_learn ""
for v in LEPNOP3 LNOATOMIC LEPNOVAL LEPNOW; do echo -n "lr5m_learn $v: "; run lr5m_learn $V/libnsk_$v.so; done
