#!/bin/bash
# Round 6, GPU session 26: wide quads from 400 000 variables per handle on -- the shard tests (1.25M-variable shards are
# now laid out in wide quads), the learning kernel's grid, the bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py tests/test_config4_gpu.py tests/test_multirank_gpu.py tests/test_partial_factors_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "not lr50m" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us  parity %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['parity'].get('ok')))"
}
for cap in 512 640 768 896 1024 1152 1280; do run new ising10m_learn 100 NSK_LEARN_TABW_GRID_CAP=$cap; done
for cap in 768 1024 1536 2048; do run new ising40m_learn 50 NSK_LEARN_TABW_GRID_CAP=$cap; done
for w in ising1m ising4m ising10m ising40m; do run new $w 200 X=1; done
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two ranks one device: %.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
