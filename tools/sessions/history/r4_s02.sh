#!/bin/bash
# round 4, session 2: quad-scheme table kernel -- parity, A/B against the pair kernel (variant OLD), grid sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q > gpurun_out/s02_multirank.log 2>&1; echo "multirank rc $?"; tail -3 gpurun_out/s02_multirank.log
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py -m gpu -x -q > gpurun_out/s02_parity.log 2>&1; echo "parity rc $?"; tail -5 gpurun_out/s02_parity.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/ab_lib.sh "ising10m ising40m ising1m ising10m_learn lr5m_learn" 100
for cap in -1024 -1280 -1536 -1632 -1792 -2048; do
  echo -n "quads cap $cap: "
  NSK_DIAG=1 NSK_TAB_GRID_CAP=$cap python bench.py --workload ising10m --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
done
for cap in -1792 -2048; do
  echo -n "40m quads cap $cap: "
  NSK_DIAG=1 NSK_TAB_GRID_CAP=$cap python bench.py --workload ising40m --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
done
