#!/bin/bash
# Round 6, GPU session 8: NQ quads per trip (1 / 2 / 4): the loads of a trip are issued together, so only the first
# quad's loads queue behind the previous trip's stores (vmcnt retires in order on gfx9-family parts).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in NQ1 new NQ4; do run $v ising10m 200 X=1; run $v ising40m 100 X=1; run $v ising1m 400 X=1; done
for v in new NQ4; do for cap in 768 1024 1280 1536 2048; do run $v ising10m 200 NSK_TABW_GRID_CAP=$cap; done; done
