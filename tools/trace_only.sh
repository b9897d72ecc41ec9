#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats of one bench workload, top kernels printed.
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-lr5m}
OUT=$R/gpurun_out/trace_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --steps ${2:-20} --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %7s total_ms %10.3f avg_us %9.2f pct %s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -type f -size +4M -delete
