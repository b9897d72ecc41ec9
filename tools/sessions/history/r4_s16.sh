#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s16_trace -- python3 $R/bench.py --workload boolw4m_learn --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $R/gpurun_out/s16_bench.log 2>&1
cd $R
NSK_VERBOSE=1 python bench.py --workload boolw4m_learn --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | grep "weights with one factor"
f=$(find gpurun_out/s16_trace -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print("  %-60s calls %6s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/s16_trace -type f -size +1M -delete
