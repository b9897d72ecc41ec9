#!/bin/bash
# tools/gpu.sh LOGFILE TIMEOUT COMMAND...: one gpurun call in the background; returns as soon as the
# snapshot of the repository has been pushed to the GPU box (the tree may be edited again from then on)
log=$1; shift; to=$1; shift
rm -f "$log"
( /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1 ) &
for i in $(seq 1 2400); do
  grep -q "push .* in\|status=\|refused\|no box" "$log" 2>/dev/null && break
  grep -q "sending /root/repo" "$log" 2>/dev/null && { sleep 20; break; }
  sleep 2
done
head -3 "$log"
