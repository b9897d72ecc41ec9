#!/bin/bash
# Round 5, GPU session 14 (runs ON THE GPU BOX): partial factors and the two-process runs again; the learning exchange
# of the LR graphs' 8 shards with the weight-delta push and the slice reduction in launches of their own.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s14; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_partial_factors_gpu.py tests/test_multirank_gpu.py -m gpu -q > $OUT/pf.log 2>&1
echo "pf + multirank rc $? $(tail -1 $OUT/pf.log)"
timeout 2400 python -m pytest tests/test_config5_shards_gpu.py -m gpu -q -k "lr5m or lr50m" > $OUT/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_lr*.json $OUT/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/config5_shards_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d.get("exchange_fraction"), {k: round(v["mean"], 1) for k, v in d["per_shard_us"].items()})
PY
