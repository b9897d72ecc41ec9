#!/bin/bash
# Round 6, GPU session 6: what the stages of the wide kernel cost (variants with one stage removed: wrong samples by
# construction, they price the stages), 10M and 40M grids, and traces for the kernel time itself.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"
}
for v in new W_NOPHILOX W_NOLOAD W_NOSTORE W_ALL; do run $v ising10m 200 X=1; run $v ising40m 100 X=1; run $v ising1m 400 X=1; done
