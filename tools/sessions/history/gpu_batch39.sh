#!/bin/bash
# learning table kernel: round keys recomputed per call + 8 waves per SIMD forced (variant LEANKEYS)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/numbskull_amd/variants
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%.3e updates/s  %.2f us/class" % (d["value"], d["roofline"]["avg_launch_us"]))'
run() { python bench.py --workload ising10m_learn --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt"; }
export NSK_DIAG=1
echo -n "full cap 1536: "; NSK_LEARN_GRID_CAP=1536 run
for c in 1536 1792 2048; do echo -n "LEANKEYS cap $c: "; NSK_LIB=$V/libnsk_LEANKEYS.so NSK_LEARN_GRID_CAP=$c run; done
