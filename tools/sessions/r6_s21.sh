#!/bin/bash
# Round 6, GPU session 21: what the descriptor round trip in front of a wave's first values costs the wide-quad kernel
# (ablation: the first trip's values requested from guessed bases before the descriptor is looked at; wrong samples).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_LIB=$R/numbskull_amd/variants/libnsk_OUTLINE.so timeout 900 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))"
}
for v in new OUTLINE NOFALLBACK new OUTLINE; do run $v ising10m 200 X=1; done
for v in new OUTLINE; do run $v ising40m 100 X=1; run $v ising100m 40 X=1; done
