#!/bin/bash
# Round 6, GPU session 30: the driver's flags (--steps 20 --warmup 5): what the per-call fixed cost is made of.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  echo -n "$2 steps $3 ${@:4} : "
  env NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['repeats']; print('%.4e updates/s  %.2f us/sweep (min %.2f max %.2f)  event launch avg %.2f us' % (d['value'], d['ms_per_step']*1e3, r['ms_per_step_min']*1e3, r['ms_per_step_max']*1e3, d['roofline']['avg_launch_us']))"
}
for st in 20 50 200; do run new ising10m $st X=1; done
run new ising10m 20 NSK_NO_PACK_TALLY=1
run new ising10m 200 NSK_NO_PACK_TALLY=1
