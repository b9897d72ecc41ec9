#!/bin/bash
# round 4, session 35: the inference profile of the weighted boolean graph at the last layout, then its two
# bench lines again so that they quote the traffic measured at this layout
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/profiles_r4
NSK_PROFILE_STEPS=50 bash tools/profile_gpu.sh boolw4m > /dev/null 2>&1
P=gpurun_out/prof_boolw4m
cp $P/summary.txt gpurun_out/profiles_r4/r4_boolw4m_summary.txt; cp $P/summary.json gpurun_out/profiles_r4/r4_boolw4m_summary.json
f=$(find $P/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/profiles_r4/r4_boolw4m_kernel_stats.csv
cp $P/traffic_boolw4m.json $P/issue_boolw4m.json gpurun_out/profiles_r4/
grep "dominant kernel" $P/summary.txt
python - <<PY
import json
for name in ("traffic", "issue"):
    t = json.load(open("profiles/%s.json" % name)); t.update(json.load(open("gpurun_out/profiles_r4/%s_boolw4m.json" % name)))
    json.dump(t, open("profiles/%s.json" % name, "w"), indent=1)
PY
for WL in boolw4m boolw4m_learn; do
  python bench.py --workload $WL --steps 100 --warmup 10 --no-extra > gpurun_out/profiles_r4/r4_${WL}_bench.json 2>/dev/null
  echo "bench $WL rc $? $(python -c "import json; d=json.load(open('gpurun_out/profiles_r4/r4_${WL}_bench.json')); print('%.4e' % d['value'], d['roofline']['traffic'])")"
done
find gpurun_out/prof_boolw4m -type f -size +1M -delete
