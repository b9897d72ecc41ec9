#!/bin/bash
# Round 6, GPU session 15: the wide learning kernel -- parity (small grids with the bound lowered, the 10M grid of
# config #3 at its default), learning benches against round 5, grid caps; the marginal-unit tie tests.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -3
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in new R5; do run $v ising10m_learn 100 X=1; done
run new ising10m_learn 100 NSK_NO_WIDE_LEARN=1
for cap in 768 1024 1280 1536; do run new ising10m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap; done
run new ising10m 200 X=1
timeout 1200 python -m pytest tests/test_learning_tie_gpu.py -m gpu -x -q -k "marginals_under" -s 2>&1 | grep -E "tie:|passed|failed|Error" | head
