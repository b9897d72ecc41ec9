#!/bin/bash
# Round 5, GPU session 24 (runs ON THE GPU BOX; the library of commit b21236e -- the split is not in the tree since): a colour's all-binary entry-parallel groups in a launch of their
# own with the two-candidate kernels (k_gibbs_ep<2>: 86 registers / 5 waves per SIMD instead of 102 / 4; k_learn_ep<2>:
# 114 / 4 instead of 152 / 3).  Parity with the split forced on small graphs (NSK_EP_SPLIT_MIN=1), then the LR lines
# with and without the split (NSK_EP_SPLIT_MIN=0) on this box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s24; rm -rf $OUT; mkdir -p $OUT
NSK_DIAG=1 NSK_EP_SPLIT_MIN=1 timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chromatic or general or duplicate or exercised or ghost or partition or accumulator or unpacked or edge_case or one_factor or shape" > $OUT/parity.log 2>&1
echo "parity (split forced) rc $? $(tail -1 $OUT/parity.log)"
NSK_DIAG=1 NSK_EP_SPLIT_MIN=1 timeout 900 python -m pytest tests/test_config5_shards_gpu.py tests/test_partial_factors_gpu.py tests/test_multirank_gpu.py -m gpu -x -q -k "lr5m or partial or (lr and p2plocal)" > $OUT/shards.log 2>&1
echo "LR shards (split forced) rc $? $(tail -1 $OUT/shards.log)"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch  launches/step %s' % (d['value'], d['roofline']['avg_launch_us'], d.get('launches_per_step')))"; }
for WL in lr50m lr50m_learn lr5m lr5m_learn; do
  case $WL in lr50m*) S=10; W=3; VS="split nosplit";; *) S=100; W=10; VS="split1 nosplit split1 nosplit";; esac
  for V in $VS; do
    case $V in split) unset NSK_DIAG NSK_EP_SPLIT_MIN;; split1) export NSK_DIAG=1 NSK_EP_SPLIT_MIN=1;; nosplit) export NSK_DIAG=1 NSK_EP_SPLIT_MIN=0;; esac
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps $S --warmup $W --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_DIAG NSK_EP_SPLIT_MIN
cat $OUT/bench.txt
