#!/bin/bash
# Round 6, GPU session 28: the front workgroups' chain two round trips shorter (segment and "no implicit adjacency" in the
# list entry); parity, then the bench lines against session 27's (10M 7.70 us per launch, 1M 3.35; floor 7.25 / 3.09).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "grid or config2 or 1000" 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us  parity %s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['parity'].get('ok')))"
}
for w in ising10m ising10m ising1m ising1m ising4m ising40m; do run new $w 200 X=1; done
run new ising10m_learn 100 X=1; run new ising10m_learn 100 X=1; run new ising40m_learn 50 X=1
