#!/bin/bash
# Round 6, GPU session 14: where is the crossover between the wide kernel and the round-5 one (1M / 4M grids), and what
# the padded layout does to the learning launches.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in new R5; do run $v ising4m 300 X=1; run $v ising10m_learn 100 X=1; run $v ising1m_learn 200 X=1; done
run new ising1m 400 NSK_NO_WIDE=1; run new ising4m 300 NSK_NO_WIDE=1; run new ising1m 400 NSK_NO_GRAPH=1
for WL in ising1m ising4m; do
  OUT=$R/gpurun_out/gaps_$WL; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1 )
  echo "== $WL"; python tools/trace_gaps.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) | grep -E "seg_tab|unpack|counters"
  find $OUT -type f -size +2M -delete
done
