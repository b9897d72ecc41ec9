#!/bin/bash
# Round 6, GPU session 27: are the front workgroups (the quads that are not wide) now what ends a wide launch?  Ablation:
# they return at once (wrong samples).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))"
}
for v in new NOREST new NOREST; do run $v ising10m 200 X=1; done
for v in new NOREST; do run $v ising1m 400 X=1; run $v ising40m 100 X=1; done
