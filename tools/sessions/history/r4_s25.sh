#!/bin/bash
# round 4, session 25: why the padded shape classes are slow -- kernel stats, and the word limit of a shape tile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-20} --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for mw in 16 20 24; do
  echo -n "boolw4m_learn max words $mw: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=$mw one boolw4m_learn
  echo -n "boolw4m max words $mw: "; NSK_DIAG=1 NSK_SHAPE_MAX_WORDS=$mw one boolw4m
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s25_prof -- python3 $R/bench.py --workload boolw4m_learn --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<P
import csv,glob
for f in glob.glob('$R/gpurun_out/s25_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
P
find $R/gpurun_out/s25_prof -type f -size +1M -delete
