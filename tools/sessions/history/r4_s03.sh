#!/bin/bash
# round 4, session 3: lag by fusion, resident grids from real occupancy, device-side state transfer, LR 2-rank debug
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
LAG=0 timeout 300 python tools/debug/lr2.py > gpurun_out/s03_lr2_lag0.log 2>&1; echo "lr2 lag0 rc $?"; grep -c "diffs 0 \[\] evid diffs 0 \[\] weight diffs 0" gpurun_out/s03_lr2_lag0.log; grep "sweep" gpurun_out/s03_lr2_lag0.log | head -8 | cut -c1-400
LAG=1 timeout 300 python tools/debug/lr2.py > gpurun_out/s03_lr2_lag1.log 2>&1; echo "lr2 lag1 rc $?"; grep "sweep" gpurun_out/s03_lr2_lag1.log | head -8 | cut -c1-400
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py tests/test_cabi.py -m gpu -x -q > gpurun_out/s03_parity.log 2>&1; echo "parity rc $?"; tail -4 gpurun_out/s03_parity.log
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py tests/test_config4_gpu.py -m gpu -x -q -k "lr5m or grid10m or config4" > gpurun_out/s03_shards.log 2>&1; echo "shards rc $?"; tail -3 gpurun_out/s03_shards.log
python tools/debug/xfer_time.py 2>&1 | tail -5
NSK_VARIANTS="new OLD" bash tools/ab_lib.sh "ising10m ising40m ising1m ising10m_learn" 100
for cap in -1536 -1792 -2048; do
  for w in ising10m ising40m; do
  echo -n "WPE8 $w cap $cap: "
  NSK_LIB=$R/numbskull_amd/variants/libnsk_TABWPE8.so NSK_DIAG=1 NSK_TAB_GRID_CAP=$cap python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
  done
done
for pcu in 3 4 5 6 7; do
  for w in lr5m lr5m_learn; do
  echo -n "EP_PER_CU $pcu $w: "
  NSK_DIAG=1 NSK_EP_PER_CU=$pcu python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
  done
done
