#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d  layoutB %.1f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['layout_bytes_per_update']))"; }
V=$R/numbskull_amd/variants
export NSK_DIAG=1
for gb in 1024 2048 8192 65536; do
  echo -n "lr5m NOHUB gen_block=$gb: "; NSK_GEN_BLOCK=$gb run lr5m $V/libnsk_NOHUB.so
done
echo -n "lr5m_learn NOHUB?: "; run lr5m_learn ""
