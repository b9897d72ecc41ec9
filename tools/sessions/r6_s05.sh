#!/bin/bash
# Round 6, GPU session 5: the wide kernel rewritten for few scalar instructions, against the session-3 build (S03) and
# round 5 on one box; issue-side counters of the new kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -2
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep' % (d['value'], d['ms_per_step']*1e3))"
}
for v in new S03 R5; do run $v ising10m 200 X=1; run $v ising1m 400 X=1; run $v ising40m 100 X=1; done
run new ising10m 200 X=1; run S03 ising10m 200 X=1
for cap in 1280 1536 2048; do run new ising10m 200 NSK_TABW_GRID_CAP=$cap; done
timeout 300 bash tools/pmc.sh w10m "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" --workload ising10m 2>&1 | grep -E "tabw" 
timeout 300 bash tools/pmc.sh w10m2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" --workload ising10m 2>&1 | grep -E "tabw"
timeout 300 bash tools/pmc.sh w10m3 "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --workload ising10m 2>&1 | grep -E "tabw"
