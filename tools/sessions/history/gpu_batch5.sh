#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base: "; run lr5m ""
for v in NOHUB EPNOP1 EPNOP2 NODRAW EPSMALL NOHUB+EPSMALL U1 U4 NOHUB+EPNOP1+EPNOP2; do
  echo -n "lr5m $v: "; run lr5m $V/libnsk_$v.so
done
NSK_LIB=$V/libnsk_TIMING.so python tools/timing_ep.py 5000000
