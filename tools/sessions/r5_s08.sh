#!/bin/bash
# Round 5, GPU session 8 (runs ON THE GPU BOX): the single-GPU table kernels against the round-4 library on one box
# after the learning quad scheme went out again and the default build's device allocations are round 4's sequence.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s08; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/sweep  %.2f us/launch' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['avg_launch_us']))"; }
for REP in 1 2; do
for V in R4 new; do
  if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
  for WL in ising10m ising10m_learn ising1m; do
    for ST in 20 200; do
      [ $WL != ising10m ] && [ $ST = 20 ] && continue
      echo -n "$WL $V steps $ST: " >> $OUT/bench.txt
      python bench.py --workload $WL --steps $ST --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
    done
  done
done
done
unset NSK_LIB
cat $OUT/bench.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
