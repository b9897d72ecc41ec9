#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "one_factor or boolw or learning_matches or every_factor or general_tiles or unpacked" > gpurun_out/s17_parity.log 2>&1; echo "parity rc $?"; tail -3 gpurun_out/s17_parity.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s17_trace -- python3 $R/bench.py --workload boolw4m_learn --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $R/gpurun_out/s17_bench.log 2>&1
cd $R
tail -1 gpurun_out/s17_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('boolw4m_learn %.4e updates/s  %.2f us/launch  ok %s' % (d['value'], d['roofline']['avg_launch_us'], d['parity']['ok']))"
f=$(find gpurun_out/s17_trace -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:6]:
    print("  %-60s calls %6s avg %10.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/s17_trace -type f -size +1M -delete
