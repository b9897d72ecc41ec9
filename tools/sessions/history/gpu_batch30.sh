#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --workload $1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
export NSK_DIAG=1
for b in 512 2048 4096; do echo -n "lr50m ep_block=$b: "; NSK_EP_BLOCK=$b run lr50m; done
