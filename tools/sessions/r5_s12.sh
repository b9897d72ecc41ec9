#!/bin/bash
# Round 5, GPU session 12 (runs ON THE GPU BOX): the host's share of a 20-sweep timed block (the driver's flags) with
# the runtime polling its completion signals instead of sleeping on an interrupt (HSA_ENABLE_INTERRUPT=0).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s12; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/sweep (min %.2f max %.2f)  %.2f us/launch' % (d['value'], d['ms_per_step'] * 1e3, d['repeats']['ms_per_step_min'] * 1e3, d['repeats']['ms_per_step_max'] * 1e3, d['roofline']['avg_launch_us']))"; }
for REP in 1 2 3; do
for MODE in irq poll; do
  if [ $MODE = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  for WL in ising10m ising1m; do
    echo -n "$WL $MODE: " >> $OUT/bench.txt
    python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$MODE.err | line >> $OUT/bench.txt
  done
done
done
cat $OUT/bench.txt
