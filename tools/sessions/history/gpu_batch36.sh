#!/bin/bash
# after the table-kernel grid change: partial profile collection (the grid workloads), then a
# diagnostic sweep of the entry-parallel kernels' workgroups per CU (NSK_EP_PER_CU)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_PROFILE_PARTIAL=1 NSK_PROFILE_WORKLOADS="ising10m ising10m_learn ising1m" NSK_PROFILE_BENCH_ONLY="ising40m" bash tools/collect_profiles.sh 2>&1 | grep -v "^r3_\|^traffic"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%.3e updates/s  %.2f us/launch" % (d["value"], d["roofline"]["avg_launch_us"]))'
export NSK_DIAG=1
for w in lr5m lr5m_learn; do for c in 3 4 5 6 7; do
  [ $w = lr5m_learn ] && [ $c -gt 5 ] && continue
  echo -n "$w workgroups per CU $c: "; NSK_EP_PER_CU=$c python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "$fmt"; done; done
