#!/bin/bash
# Round 5, GPU session 16 (runs ON THE GPU BOX): the closing weight gather of a large exchange as a one-wave wait +
# a many-block copy; the graph compiler's word cache at 50M (lap table of the lr50m_learn bench run).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s16; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "large or (lr and learn and p2p)" > $OUT/multirank.log 2>&1
echo "multirank rc $? $(tail -1 $OUT/multirank.log)"
timeout 1500 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "lr5m or lr50m" > $OUT/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_lr*.json $OUT/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/config5_shards_lr*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["exchange_fraction"], {k: round(v["mean"], 1) for k, v in d["per_shard_us"].items()})
PY
NSK_VERBOSE=1 python bench.py --workload lr50m_learn --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/lr50m_learn_bench.json 2> $OUT/lr50m_learn.err
echo "lr50m_learn rc $?"
grep "compile" $OUT/lr50m_learn.err | grep -v colour > $OUT/lr50m_learn_compile_laps.txt; cat $OUT/lr50m_learn_compile_laps.txt
python -c "
import json; d=json.loads(open('$OUT/lr50m_learn_bench.json').read().strip().splitlines()[-1]); print('%.4e' % d['value'], d['config'].get('generate_s'), d['config'].get('load_and_compile_s'), d['config'].get('compile_s'))"
timeout 900 python -m pytest tests/test_config5_gpu.py -m gpu -x -q > $OUT/config5.log 2>&1
echo "config5 rc $? $(tail -1 $OUT/config5.log)"
