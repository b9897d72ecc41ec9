#!/bin/bash
# Round 5, GPU session 21 (runs ON THE GPU BOX): the rows of an entry-parallel pass two deep in flight (ep_pass_deep,
# -DNSK_EP_DEEP=1: libnsk_DEEP.so) and two rows per step (-DNSK_EP_U_INF=2: libnsk_EPU2.so) against the tree's library
# (one row per step, one deep) -- parity of the DEEP library first, then the inference lines on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s21; rm -rf $OUT; mkdir -p $OUT
NSK_LIB=$R/numbskull_amd/variants/libnsk_DEEP.so timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chromatic or general or duplicate or exercised or ghost or partition or accumulator or unpacked or edge_case or one_factor or shape" > $OUT/parity.log 2>&1
echo "parity (DEEP) rc $? $(tail -1 $OUT/parity.log)"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in lr5m boolw4m lr50m; do
  case $WL in lr50m*) VS="new DEEP EPU2"; S=10; W=3;; *) VS="new DEEP EPU2 new DEEP EPU2"; S=100; W=10;; esac
  for V in $VS; do
    if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps $S --warmup $W --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_LIB
cat $OUT/bench.txt
