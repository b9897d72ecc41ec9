#!/bin/bash
# Round 5, GPU session 10 (runs ON THE GPU BOX): the draw table carried in a vector register (ds_bpermute lookups)
# against the table read from memory (variant NOLT), at a good and at a bad placement of the table (NSK_ALLOC_PAD=2).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s10; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/sweep  %.2f us/launch' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['avg_launch_us']))"; }
export NSK_DIAG=1
for REP in 1 2; do
for V in new NOLT; do
  if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
  for PAD in 0 2; do
    export NSK_ALLOC_PAD=$PAD
    for WL in ising10m ising40m ising1m; do
      echo -n "$WL $V pad$PAD: " >> $OUT/bench.txt
      python bench.py --workload $WL --steps 200 --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
    done
  done
done
done
unset NSK_LIB NSK_ALLOC_PAD NSK_DIAG
cat $OUT/bench.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
