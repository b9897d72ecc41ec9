#!/bin/bash
# Round 6, GPU session 2: diagnosis of the first wide-quad run (1M grid twice as slow, 40M grid 200x).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/r6_s02; O=gpurun_out/r6_s02
timeout 600 python tools/debug/wide_diag.py parity 2>&1 | tail -8
for WL in ising1m ising10m; do
for V in "" "NSK_NO_WIDE=1" "NSK_NO_GRAPH=1"; do
  env NSK_DIAG=1 $V timeout 300 python bench.py --steps 20 --warmup 5 --workload $WL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', '$V', d['value'], d['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof1m -o p -- python3 $R/bench.py --steps 20 --warmup 5 --workload ising1m > /dev/null 2>&1
cd $R; f=$(find $O/prof1m -name "*kernel_stats.csv" | head -1); head -8 $f
