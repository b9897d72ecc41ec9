#!/bin/bash
# Round 6, GPU session 31: 20-step blocks after 15 / 100 / 500 / 2000 untimed sweeps (does the device need longer to reach its clocks?).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in 5 100 500 2000 5; do
  echo -n "warmup $w : "
  timeout 300 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['repeats']; print('%.4e updates/s  %.2f us/sweep (min %.2f max %.2f)' % (d['value'], d['ms_per_step']*1e3, r['ms_per_step_min']*1e3, r['ms_per_step_max']*1e3))"
done
