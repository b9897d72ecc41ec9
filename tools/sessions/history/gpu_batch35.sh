#!/bin/bash
# inference table kernel: grid-cap sweep, second pass (repeats; 40M and 1M grids)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%.3e updates/s  %.2f us/class" % (d["value"], d["roofline"]["avg_launch_us"]))'
export NSK_DIAG=1
for rep in 1 2; do for c in 1536 1792 2048; do echo -n "ising10m grid cap $c: "; NSK_TAB_GRID_CAP=$c python bench.py --workload ising10m --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "$fmt"; done; done
for c in 1280 1536 1792 2048; do echo -n "ising40m grid cap $c: "; NSK_TAB_GRID_CAP=$c python bench.py --workload ising40m --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "$fmt"; done
for c in 512 1024 1536 2048; do echo -n "ising1m grid cap $c: "; NSK_TAB_GRID_CAP=$c python bench.py --workload ising1m --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "$fmt"; done
