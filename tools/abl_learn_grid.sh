#!/bin/bash
# prices the stages of the learning table kernel on the 10M / 1M grids (variants from
# NSK_ABL_TU=learn tools/build_ablations.sh NOPHILOX)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in ising10m_learn ising1m_learn; do
for v in ${NSK_VARIANTS:-full NOPHILOX}; do
  lib=""; [ "$v" != full ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "$w variant=$v "
  NSK_LIB=$lib python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/class' % (d['value'], d['roofline']['avg_launch_us']))"
done; done
