#!/bin/bash
# Round 6, GPU session 29: instruction fetch at the start of a launch -- s_nop instructions (256 / 1024 / 4096: 1 / 4 / 16 KB
# of straight-line code, 0.12 / 0.5 / 2 us to execute) in front of everything in the wide-quad kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  launch %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))"
}
for v in new NOPS256 NOPS1024 NOPS4096 new; do run $v ising10m 200 X=1; done
for v in new NOPS256 NOPS1024 NOPS4096; do run $v ising1m 400 X=1; done
