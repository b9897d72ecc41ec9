#!/bin/bash
# Round 5, GPU session 13 (runs ON THE GPU BOX): the GPU tests of the round's last additions -- a shuffled grid
# repartitioned by numbskull_amd.partition and sampled in 8 shards, partial factors on 4 shards (voter graph, LR graph;
# inference and learning) -- and the two-process runs once more (PartitionedSampler gathers partial-factor requests now).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s13; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_partial_factors_gpu.py tests/test_multirank_gpu.py -m gpu -x -q > $OUT/pf.log 2>&1
echo "pf + multirank rc $? $(tail -1 $OUT/pf.log)"
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "shuffled or fused" > $OUT/shuffled.log 2>&1
echo "shuffled rc $? $(tail -1 $OUT/shuffled.log)"
cp gpurun_out/config5_shards_shuffled1m_inference.json $OUT/ 2>/dev/null
tail -30 $OUT/pf.log | grep -v "^$" | head -40
