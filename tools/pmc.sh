#!/bin/bash
# Usage (on the GPU box): tools/pmc.sh TAG "COUNTER1 COUNTER2 ..." [bench args]
# One rocprofv3 --pmc pass of bench.py; prints per-kernel means.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
CTRS=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        res[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in res.items():
    if "gibbs" in k or "learn" in k:
        for cn,v in sorted(c.items()):
            print("%-50s %-26s %.5g (n=%d)"%(k,cn,sum(v)/len(v),len(v)))
PY
