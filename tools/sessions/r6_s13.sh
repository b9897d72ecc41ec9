#!/bin/bash
# Round 6, GPU session 13: where does the 1M grid's launch period go?  Kernel durations and gaps (kernel trace), new vs R5.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in new R5; do
  lib=""; [ "$v" != new ] && lib="$R/numbskull_amd/variants/libnsk_$v.so"
  for e in X=1 NSK_NO_GRAPH=1; do
    OUT=$R/gpurun_out/gaps_${v}_$e; rm -rf $OUT; mkdir -p $OUT
    ( cd /tmp; export TMPDIR=/tmp; env NSK_LIB=$lib NSK_DIAG=1 $e rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload ising1m --steps 200 --warmup 20 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1 )
    echo "== $v $e: $(tail -1 $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/sweep' % (d['ms_per_step']*1e3))")"
    python tools/trace_gaps.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) | grep -E "seg_tab|unpack|counters"
    find $OUT -type f -size +2M -delete
  done
done
