#!/bin/bash
# Round 5, GPU session 30 (runs ON THE GPU BOX): k_learn_ep at four waves per SIMD as the tree's default -- parity
# of everything that learns through it, then the three learning bench lines (partial collection).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT0=$R/gpurun_out/r5_s30; rm -rf $OUT0; mkdir -p $OUT0
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $OUT0/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT0/parity.log)"
timeout 600 python -m pytest tests/test_config5_shards_gpu.py tests/test_partial_factors_gpu.py tests/test_multirank_gpu.py -m gpu -x -q -k "(lr5m and True) or partial or (lr and learn)" > $OUT0/shards.log 2>&1
echo "LR shards rc $? $(tail -1 $OUT0/shards.log)"
mkdir -p $OUT0/keep; cp gpurun_out/config5_shards_lr5m_*.json $OUT0/keep/ 2>/dev/null
NSK_PROFILE_STAGE=bench NSK_PROFILE_PARTIAL=1 NSK_PROFILE_SKIP_DEFAULT=1 NSK_PROFILE_BENCH_WORKLOADS="lr5m_learn boolw4m_learn" \
  NSK_PROFILE_BENCH_ONLY="lr50m_learn" bash -c 'sed "s/^timeout 2400 python -m pytest tests\/test_config5_shards_gpu.py.*$/true/" tools/collect_profiles.sh > /tmp/collect_nolast.sh; bash /tmp/collect_nolast.sh' > $OUT0/collect.log 2>&1
OUT=$R/gpurun_out/profiles_r5
python - <<PY
import json
for n in ("lr5m_learn", "boolw4m_learn", "lr50m_learn"):
    d = json.loads(open("$OUT/r5_%s_bench.json" % n).read().strip().splitlines()[-1])
    print(n, "%.4e" % d["value"], round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_us"], 1))
PY
