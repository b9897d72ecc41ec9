#!/bin/bash
# Round 6, GPU session 4: the wide kernel with value loads one trip ahead and the table fill behind the first requests;
# variants D1 (descriptor ahead only), S96 / S80 (scalar registers capped: 7 / 8 blocks per CU), grid caps; vs round 5.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"
}
for v in new D1 R5; do run $v ising10m 200 X=1; run $v ising1m 400 X=1; done
for v in new S96 S80; do for cap in 1536 1792 2048; do run $v ising10m 200 NSK_TABW_GRID_CAP=$cap; done; done
for v in new S80; do run $v ising40m 100 NSK_TABW_GRID_CAP=2048; run $v ising1m 400 X=1; done
run new ising40m 100 X=1; run new ising100m 50 X=1; run R5 ising100m 50 X=1
