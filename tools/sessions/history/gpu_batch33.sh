#!/bin/bash
# learning table kernel: grid-cap sweep around the resident capacity, 1M grid check
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/numbskull_amd/variants
run() { python bench.py --workload $1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"; }
export NSK_DIAG=1
for c in 1280 1408 1536 1664 1792 1920 2048; do echo -n "ising10m_learn grid cap $c: "; NSK_LEARN_GRID_CAP=$c run ising10m_learn; done
for c in 512 1024 1536 2048; do echo -n "ising1m_learn grid cap $c: "; NSK_LEARN_GRID_CAP=$c run ising1m_learn; done
unset NSK_DIAG
echo -n "ising1m_learn BASE: "; NSK_LIB=$V/libnsk_BASE.so run ising1m_learn
echo -n "ising1m_learn full: "; run ising1m_learn
