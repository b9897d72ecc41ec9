#!/bin/bash
# Round 6, GPU session 16: grid of the wide learning kernel (k_learn_seg_tabw) on the 10M / 40M grids.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for cap in 1408 1536 1664 1792 2048 2560 3072 4096 8192; do run new ising10m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap; done
run new ising10m_learn 100 NSK_NO_WIDE_LEARN=1
for cap in 1536 2048 4096; do run new ising40m_learn 50 NSK_LEARN_TABW_GRID_CAP=$cap; done
run new ising40m_learn 50 NSK_NO_WIDE_LEARN=1
for cap in 1536 2048; do run new ising4m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap NSK_WIDE_MIN=0; done
run new ising4m_learn 100 NSK_NO_WIDE_LEARN=1
