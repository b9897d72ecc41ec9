#!/bin/bash
# inference table kernel: grid-cap sweep (NSK_TAB_GRID_CAP, diagnostic)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"; }
export NSK_DIAG=1
for c in 1024 1280 1536 1792 2048; do echo -n "ising10m grid cap $c: "; NSK_TAB_GRID_CAP=$c run ising10m; done
for c in 1024 1536 2048; do echo -n "ising40m grid cap $c: "; NSK_TAB_GRID_CAP=$c python bench.py --workload ising40m --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"; done
unset NSK_DIAG
echo -n "ising10m_learn default (1536): "; python bench.py --workload ising10m_learn --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"
