#!/bin/bash
# Round 6, GPU session 23: the quads of a wide launch that are not wide ones sampled by workgroups of their own in front
# of the grid (NSK_DIAG=1 NSK_NO_TABW_REST=1: in line, as before).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us  parity %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['parity'].get('ok')))"
}
for e in X=1 NSK_NO_TABW_REST=1 X=1 NSK_NO_TABW_REST=1; do run new ising10m 200 $e; done
for e in X=1 NSK_NO_TABW_REST=1; do run new ising40m 100 $e; run new ising100m 40 $e; run new ising4m 200 $e NSK_WIDE_MIN=0; done
for cap in 1280 1536 1792 2048; do run new ising10m 200 NSK_TABW_GRID_CAP=$cap; done
