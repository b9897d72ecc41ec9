#!/bin/bash
# Round 6, GPU session 12: wide kernel v3 -- hot arguments preloaded into scalar registers, per-wave LDS table, cold
# arguments behind the first requests.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -3
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in new NQ1 R5; do run $v ising10m 200 X=1; run $v ising1m 400 X=1; run $v ising40m 100 X=1; done
for cap in 1024 1536 2048; do run new ising10m 200 NSK_TABW_GRID_CAP=$cap; done
echo "=== TIMING 10M"; NSK_LIB=$R/numbskull_amd/variants/libnsk_TIMING.so timeout 200 python tools/timing_tabw.py 2500 4000 2>&1 | grep -E "entry ->|landed|trips, each"
echo "=== TIMING 1M"; NSK_LIB=$R/numbskull_amd/variants/libnsk_TIMING.so timeout 200 python tools/timing_tabw.py 1000 1000 2>&1 | grep -E "entry ->|landed|trips, each"
