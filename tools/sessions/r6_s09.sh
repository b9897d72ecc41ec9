#!/bin/bash
# Round 6, GPU session 9: why two quads per trip are slower than one -- stage ablations on NQ = 2 and issue counters of both.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"
}
for v in new W2_NOPHILOX W2_NOLOAD W2_NOSTORE NQ1; do run $v ising10m 200 X=1; done
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"
C3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
for lib in "" "$R/numbskull_amd/variants/libnsk_NQ1.so"; do
  echo "== PMC lib=[$lib]"
  NSK_LIB=$lib timeout 300 bash tools/pmc.sh a "$C1" --workload ising10m 2>&1 | grep -E "tabw"
  NSK_LIB=$lib timeout 300 bash tools/pmc.sh b "$C2" --workload ising10m 2>&1 | grep -E "tabw"
  NSK_LIB=$lib timeout 300 bash tools/pmc.sh c "$C3" --workload ising10m 2>&1 | grep -E "tabw"
done
