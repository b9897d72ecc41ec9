#!/bin/bash
# Round 5, GPU session 31 (runs ON THE GPU BOX): the four-waves cap on the eight-candidate learning kernel only
# (k_learn_ep_w4; the two-candidate k_learn_ep without it, as before) -- parity of the learning paths, three lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT0=$R/gpurun_out/r5_s31; rm -rf $OUT0; mkdir -p $OUT0
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "learn or chromatic or general or accumulator or one_factor or shape" > $OUT0/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT0/parity.log)"
timeout 300 python -m pytest tests/test_config5_shards_gpu.py tests/test_multirank_gpu.py -m gpu -x -q -k "(lr5m and True) or (lr and learn and p2plocal)" > $OUT0/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT0/shards.log)"
for WL in lr5m_learn boolw4m_learn; do
  python bench.py --workload $WL --steps 100 --warmup 10 --no-extra > $OUT0/r5_${WL}_bench.json 2> /dev/null
  python -c "import json; d=json.loads(open('$OUT0/r5_${WL}_bench.json').read().strip().splitlines()[-1]); print('$WL %.4e %.1f us/launch %s' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['kernel']))"
done
