#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "persist8 "; NSK_PERSIST=8 run
echo -n "persist8 PAIR (timing only) "; NSK_PERSIST=8 NSK_LIB=$R/numbskull_amd/variants/libnsk_PAIR.so run
echo -n "persist16 PAIR (timing only) "; NSK_PERSIST=16 NSK_LIB=$R/numbskull_amd/variants/libnsk_PAIR.so run
echo -n "persist6 PAIR (timing only) "; NSK_PERSIST=6 NSK_LIB=$R/numbskull_amd/variants/libnsk_PAIR.so run
