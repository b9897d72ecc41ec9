#!/bin/bash
# Round 6, GPU session 25: the learning launches' plans prepared once per handle; from which size wide quads pay now
# (inference and learning); the learning kernel's grid.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us  parity %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['parity'].get('ok')))"
}
for e in X=1 NSK_NO_TABW_REST=1 NSK_NO_WIDE_LEARN=1 X=1; do run new ising10m_learn 100 $e; done
for cap in 1024 1536 2048; do run new ising10m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap; done
for e in X=1 NSK_NO_TABW_REST=1 NSK_NO_WIDE_LEARN=1; do run new ising40m_learn 50 $e; done
for w in ising4m_learn ising1m_learn; do run new $w 200 NSK_WIDE_MIN=99999999; run new $w 200 NSK_WIDE_MIN=0 NSK_WIDE_LEARN_MIN=0;  run new $w 200 NSK_WIDE_MIN=0 NSK_WIDE_LEARN_MIN=999999; done
for w in ising64k ising256k ising500k ising1m ising4m; do run new $w 400 NSK_WIDE_MIN=99999999; run new $w 400 NSK_WIDE_MIN=0; done
