#!/bin/bash
# round 4, session 21: the internal numbering of single-factor weights (nsk_compile.h wmap): parity of the
# graphs it touches, then the weighted boolean graph with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "one_factor or weight_slots or boolean or direct or capture or lag" 2>&1 | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
one() { python bench.py --workload $1 --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn caller's numbering: "; NSK_DIAG=1 NSK_NO_WORDER=1 one boolw4m_learn
echo -n "boolw4m: "; one boolw4m
echo -n "boolw4m caller's numbering: "; NSK_DIAG=1 NSK_NO_WORDER=1 one boolw4m
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s21_prof -o bl -- python3 $R/bench.py --workload boolw4m_learn --steps 30 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<P
import csv,glob
for f in glob.glob('$R/gpurun_out/s21_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
P
