#!/bin/bash
# On the GPU box: time the lr5m inference sweep with each ablated library (tools/build_ablations.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" NODRAW NOGATHER NOLUT NOWALK "$@"; do
  lib=""; [ -n "$v" ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "variant=${v:-full} "
  NSK_LIB=$lib python bench.py --workload lr5m --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
done
