#!/bin/bash
# XCD-private bins (close_sink): parity of the learning paths, then timings and the grid-cap sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/numbskull_amd/variants
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py -m gpu -x -q -k "learn or chromatic or planted" > gpurun_out/b32_main.log 2>&1; echo "learn parity rc $?"; tail -2 gpurun_out/b32_main.log
run() { python bench.py --workload $1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"; }
for v in full BASE NOAPPLY LNOSINK NOPHILOX+LNOEV+LNOSINK+LNOBALLOT+LNOINIT+LNOEVST; do
  lib=""; [ "$v" != full ] && lib="$V/libnsk_$v.so"
  echo -n "ising10m_learn variant=$v "; NSK_LIB=$lib run ising10m_learn
done
export NSK_DIAG=1
for c in 1024 1536 3072 4096; do echo -n "ising10m_learn grid cap $c: "; NSK_LEARN_GRID_CAP=$c run ising10m_learn; done
unset NSK_DIAG
echo -n "ising1m_learn full: "; run ising1m_learn
echo -n "ising1m_learn BASE: "; NSK_LIB=$V/libnsk_BASE.so run ising1m_learn
