#!/bin/bash
# Round 5, GPU session 28 (runs ON THE GPU BOX): closing partial collection at the library with the scalar wave index
# in the entry-parallel passes -- the bench lines of the workloads that take those kernels, the whole GPU suite
# (8-shard runs included) and the smoke run.  (The rocprofv3 passes of those workloads stay those of session 22: the
# kernels differ by the scalar index, 1-3 % in time, not in bytes.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_PROFILE_STAGE=bench NSK_PROFILE_PARTIAL=1 NSK_PROFILE_SKIP_DEFAULT=1 NSK_PROFILE_BENCH_WORKLOADS="lr5m lr5m_learn boolw4m boolw4m_learn" \
  NSK_PROFILE_BENCH_ONLY="lr50m lr50m_learn" NSK_PROFILE_FULL_TESTS=1 bash tools/collect_profiles.sh
OUT=$R/gpurun_out/profiles_r5
python - <<PY
import json, glob
for n in ("lr5m", "lr5m_learn", "boolw4m", "boolw4m_learn", "lr50m", "lr50m_learn"):
    d = json.loads(open("$OUT/r5_%s_bench.json" % n).read().strip().splitlines()[-1])
    print(n, "%.4e" % d["value"], round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_us"], 1), d["config"].get("compile_s"))
PY
tail -3 $OUT/r5_gpu_tests.log
