#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches']))"; }
V=$R/numbskull_amd/variants
for v in NOHUB NOHUB+EPNOW NOHUB+EPNOVAL NOHUB+EPNOW+EPNOVAL; do
  echo -n "lr5m $v: "; run lr5m $V/libnsk_$v.so
done
cd /tmp; export TMPDIR=/tmp
for c in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | cut -c1-12 | tr ' ' '_')
  $R/tools/pmc.sh ep_$tag "$c" --workload lr5m 2>&1 | grep -E "k_gibbs_ep" 
done
