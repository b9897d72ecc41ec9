#!/bin/bash
# copy the newest kernel-trace stats of tools/trace_only.sh runs into profiles/ (run here, after gpurun)
for WL in "$@"; do
  f=$(ls -t gpurun_out/trace_$WL/trace/*/*_kernel_stats.csv | head -1)
  cp "$f" profiles/r2_${WL}_kernel_stats.csv
  grep -h '"metric"' gpurun_out/trace_$WL/bench.log > profiles/r2_${WL}_bench.json
done
