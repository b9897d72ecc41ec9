#!/bin/bash
# Runs ON THE GPU BOX: per-dispatch durations and gaps of one bench workload (kernel trace), printed per
# position of the launch inside the sweep.  usage: tools/trace_dispatches.sh WORKLOAD [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-lr5m}
OUT=$R/gpurun_out/disp_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --steps ${2:-4} --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gibbs" in r["Kernel_Name"] or "learn" in r["Kernel_Name"] or "apply" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-200:]
prev = None
for r in rows[-48:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s grid %8s wg %4s lds %6s vgpr %4s dur %8.1f us gap %7.1f us" % (r["Kernel_Name"].split("(")[0][10:54], r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
PY
find $OUT -type f -size +4M -delete
