#!/bin/bash
# Round 5, GPU session 33 (runs ON THE GPU BOX): the driver's own commands at the round's last library -- smoke,
# `bench.py --steps 20 --warmup 5`, `bench.py`.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT0=$R/gpurun_out/r5_s33; rm -rf $OUT0; mkdir -p $OUT0
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT0/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $OUT0/smoke.log)"
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT0/driver_flags_bench.json 2> $OUT0/err1.txt; echo "driver flags rc $?"
python -c "
import json; d=json.loads(open('$OUT0/driver_flags_bench.json').read().strip().splitlines()[-1]); print('%.4e' % d['value'], round(d['ms_per_step']*1e3,2), 'us/sweep', d['parity'].get('statistics'), {k: '%.3e' % v['value'] for k, v in d['also'].items()})"
