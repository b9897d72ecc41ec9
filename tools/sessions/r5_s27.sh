#!/bin/bash
# Round 5, GPU session 27 (runs ON THE GPU BOX): the wave's index as a scalar (readfirstlane) in the entry-parallel
# passes -- row numbers, member counts, sub-row offsets and their branches become scalar work: k_gibbs_ep<8> 102 -> 90
# vector registers (5 waves per SIMD instead of 4), row step 157 -> 145 vector instructions.  Parity, then the LR and
# weighted-boolean lines against the library before (libnsk_CNT.so) on this box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s27; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chromatic or general or duplicate or exercised or ghost or partition or accumulator or unpacked or edge_case or one_factor or shape" > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 900 python -m pytest tests/test_config5_shards_gpu.py tests/test_partial_factors_gpu.py -m gpu -x -q -k "lr5m or partial" > $OUT/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT/shards.log)"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in lr5m lr5m_learn boolw4m boolw4m_learn lr50m lr50m_learn; do
  case $WL in lr50m*) VS="new CNT"; S=10; W=3;; *) VS="new CNT new CNT"; S=100; W=10;; esac
  for V in $VS; do
    if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps $S --warmup $W --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_LIB
cat $OUT/bench.txt
