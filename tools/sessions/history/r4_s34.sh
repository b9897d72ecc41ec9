#!/bin/bash
# round 4, session 34: shape classes only for lists of <= 4 words where the entry-parallel groups can take the
# variable: kernel-level parity tests, the two bench lines of the weighted boolean graph, the learning profile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/profiles_r4
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x > gpurun_out/profiles_r4/r4_gpu_tests_s34.log 2>&1
echo "hip parity rc $? $(tail -1 gpurun_out/profiles_r4/r4_gpu_tests_s34.log)"
for WL in boolw4m boolw4m_learn; do
  python bench.py --workload $WL --steps 100 --warmup 10 --no-extra > gpurun_out/profiles_r4/r4_${WL}_bench.json 2>/dev/null
  echo "bench $WL rc $? $(python -c "import json; d=json.load(open('gpurun_out/profiles_r4/r4_${WL}_bench.json')); print('%.4e' % d['value'], d['parity'] if 'parity' in d else '')" | cut -c1-200)"
done
NSK_PROFILE_STEPS=50 bash tools/profile_gpu.sh boolw4m_learn > /dev/null 2>&1
P=gpurun_out/prof_boolw4m_learn
cp $P/summary.txt gpurun_out/profiles_r4/r4_boolw4m_learn_summary.txt; cp $P/summary.json gpurun_out/profiles_r4/r4_boolw4m_learn_summary.json
f=$(find $P/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/profiles_r4/r4_boolw4m_learn_kernel_stats.csv
cp $P/traffic_boolw4m_learn.json $P/issue_boolw4m_learn.json gpurun_out/profiles_r4/
grep "dominant kernel" $P/summary.txt
find gpurun_out/prof_boolw4m_learn -type f -size +1M -delete
