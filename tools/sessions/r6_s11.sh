#!/bin/bash
# Round 6, GPU session 11: where do the kernel arguments live?  The wide kernel's waves spend half their life in front of
# the first trip (tools/sessions/r6_s10.sh): dependent rounds of scalar loads from the kernarg segment.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"
}
for e in X=1 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0; do for v in new R5; do run $v ising10m 200 $e; run $v ising1m 400 $e; done; done
echo "=== TIMING 10M, device kernargs"; HIP_FORCE_DEV_KERNARG=1 NSK_LIB=$R/numbskull_amd/variants/libnsk_TIMING.so timeout 200 python tools/timing_tabw.py 2500 4000 2>&1 | grep -E "entry ->|landed|trips, each"
echo "=== TIMING 10M, host kernargs"; HIP_FORCE_DEV_KERNARG=0 NSK_LIB=$R/numbskull_amd/variants/libnsk_TIMING.so timeout 200 python tools/timing_tabw.py 2500 4000 2>&1 | grep -E "entry ->|landed|trips, each"
