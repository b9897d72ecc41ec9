#!/bin/bash
# Round 6, GPU session 3: the dedicated wide-quad kernel (descriptor one trip ahead) -- the new parity tests, A/B against
# the round-5 library on this box, kernel-trace stats.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/r6_s03; O=gpurun_out/r6_s03
timeout 600 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -4
ab() {  # workload steps
for v in new R5; do
  lib=""; [ "$v" != new ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  echo -n "$1 $v "
  NSK_LIB=$lib timeout 300 python bench.py --workload $1 --steps $2 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  %.2f us/launch frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; }
ab ising10m 200; ab ising1m 400; ab ising40m 100; ab ising10m 200
for cap in 1024 1536 1792 2048 3072 5120; do echo -n "cap $cap: "; NSK_DIAG=1 NSK_TABW_GRID_CAP=$cap timeout 200 python bench.py --workload ising10m --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"; done
timeout 300 bash tools/trace_only.sh ising10m 100 2>&1 | grep -E "k_gibbs|updates" | head -4
timeout 300 bash tools/trace_only.sh ising1m 100 2>&1 | grep -E "k_gibbs|updates" | head -4
