#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python bench.py --workload lr5m --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch colors %d' % (d['value'], d['roofline']['avg_launch_us'], d['config']['colors']))"; }
echo -n "NOHUB "; NSK_LIB=$R/numbskull_amd/variants/libnsk_NOHUB.so run
echo -n "MAXE24 "; NSK_GEN_MAX_ENTRIES=24 run
echo -n "MAXE24+NOHUB "; NSK_GEN_MAX_ENTRIES=24 NSK_LIB=$R/numbskull_amd/variants/libnsk_NOHUB.so run
echo -n "MAXE12 "; NSK_GEN_MAX_ENTRIES=12 run
echo -n "GENBLOCK64k "; NSK_GEN_BLOCK=65536 run
echo -n "GENBLOCK1M "; NSK_GEN_BLOCK=1048576 run
for t in 1 2 4; do echo -n "ising10m TPW=$t "; NSK_TPW=$t python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e %.2f us' % (d['value'], d['roofline']['avg_launch_us']))"; done
