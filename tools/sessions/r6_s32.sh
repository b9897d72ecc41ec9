#!/bin/bash
# Round 6, GPU session 32: captured sweep sequences of 64 sweeps beside the 16-sweep ones (NSK_DIAG=1 NSK_NO_BIG_GRAPH=1: 16 only).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_wide_quads_gpu.py tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "grid10m" 2>&1 | tail -2
run() {  # variant workload steps [env...]
  echo -n "$2 ${@:4} : "
  env NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for e in X=1 NSK_NO_BIG_GRAPH=1 X=1 NSK_NO_BIG_GRAPH=1; do run new ising1m 400 $e; done
for e in X=1 NSK_NO_BIG_GRAPH=1; do run new ising256k 400 $e; run new ising64k 400 $e; done
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two ranks one device: %.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
