#!/bin/bash
# Round 6, GPU session 24: front workgroups for the quads that are not wide, now in the learning kernel too; where wide
# quads pay from with them (1M / 4M grids); the learning kernel's grid again.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us  parity %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['parity'].get('ok')))"
}
for e in X=1 NSK_NO_TABW_REST=1 X=1; do run new ising10m_learn 100 $e; done
for cap in 1024 1280 1536 1792 2048 2560; do run new ising10m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap; done
for e in X=1 NSK_NO_TABW_REST=1; do run new ising40m_learn 50 $e; done
for cap in 1536 2048 3072; do run new ising40m_learn 50 NSK_LEARN_TABW_GRID_CAP=$cap; done
run new ising4m_learn 100 X=1; run new ising4m_learn 100 NSK_WIDE_MIN=0 NSK_WIDE_LEARN_MIN=0
run new ising1m_learn 200 X=1; run new ising1m_learn 200 NSK_WIDE_MIN=0 NSK_WIDE_LEARN_MIN=0
run new ising1m 400 X=1; run new ising1m 400 NSK_WIDE_MIN=0
run new ising4m 200 X=1; run new ising4m 200 NSK_WIDE_MIN=0
for cap in 512 768 1024 1536; do run new ising4m 200 NSK_WIDE_MIN=0 NSK_TABW_GRID_CAP=$cap; run new ising1m 400 NSK_WIDE_MIN=0 NSK_TABW_GRID_CAP=$cap; done
