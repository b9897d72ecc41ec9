#!/bin/bash
# Round 5, GPU session 15 (runs ON THE GPU BOX): large exchanges as many-block launches with one-wave flag kernels
# between them (k_p2p_push_big / k_p2p_wait / k_p2p_unpack_big): the two-process runs, then the 8-shard LR runs
# (5M: the <= 64-block kernels; 50M: the large path) with their phase timings.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s15; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "p2p or large" > $OUT/multirank.log 2>&1
echo "multirank rc $? $(tail -1 $OUT/multirank.log)"
NSK_DIAG=1 NSK_P2P_BIG_MIN=0 timeout 900 python -m pytest tests/test_config5_shards_gpu.py tests/test_partial_factors_gpu.py -m gpu -x -q -k "lr5m or partial" > $OUT/shards_forced.log 2>&1
echo "LR 5M shards + partial factors, large path forced rc $? $(tail -1 $OUT/shards_forced.log)"
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "lr5m or lr50m" > $OUT/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_lr*.json $OUT/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/config5_shards_lr*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["exchange_fraction"], {k: round(v["mean"], 1) for k, v in d["per_shard_us"].items()})
PY
