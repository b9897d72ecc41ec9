#!/bin/bash
# Round 5, GPU session 22 (runs ON THE GPU BOX): partial collection at the library with the counted row loads of the
# entry-parallel groups -- rocprofv3 passes and bench lines of the workloads whose inference sweep takes k_gibbs_ep
# (lr5m, boolw4m in full; lr50m light), the whole GPU suite and the smoke run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_PROFILE_PARTIAL=1 NSK_PROFILE_SKIP_DEFAULT=1 NSK_PROFILE_WORKLOADS="lr5m boolw4m" NSK_PROFILE_LIGHT_WORKLOADS="lr50m" \
  NSK_PROFILE_BENCH_WORKLOADS="lr5m boolw4m" NSK_PROFILE_BENCH_ONLY="lr50m" NSK_PROFILE_FULL_TESTS=1 bash tools/collect_profiles.sh
OUT=$R/gpurun_out/profiles_r5
python - <<PY
import json, glob
for n in ("lr5m", "boolw4m", "lr50m"):
    d = json.loads(open("$OUT/r5_%s_bench.json" % n).read().strip().splitlines()[-1])
    print(n, "%.4e" % d["value"], d.get("ms_per_step"), d["roofline"])
PY
tail -3 $OUT/r5_gpu_tests.log
