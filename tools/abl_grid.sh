#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for e in "" "NSK_NO_AFFINE=1"; do
for v in "$@"; do
  lib=""; [ "$v" != full ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "[$e] variant=$v "; env $e NSK_LIB=$lib bash -c "$(declare -f run); run"
done; done
