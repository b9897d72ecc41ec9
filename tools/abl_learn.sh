#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in full NOATOMIC NOPASS2; do
  lib=""; [ "$v" != full ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "lr5m_learn variant=$v "
  NSK_LIB=$lib python bench.py --workload lr5m_learn --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
done
