#!/bin/bash
# A/B of two library builds on a set of workloads: tools/ab_lib.sh "ising10m ising1m" [steps]
# (variant "OLD" = numbskull_amd/variants/libnsk_OLD.so, saved before a change)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in $1; do
for v in ${NSK_VARIANTS:-new OLD}; do
  lib=""; [ "$v" != new ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "$w $v "
  NSK_LIB=$lib python bench.py --workload $w --steps ${2:-200} --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch frac %.3f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; done
