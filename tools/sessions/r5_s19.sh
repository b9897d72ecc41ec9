#!/bin/bash
# Round 5, GPU session 19 (runs ON THE GPU BOX; session 18 again after its one failure -- a rendezvous port taken by another process, tests/util.py free_port since): the 50M lines with the index build and the state arrays
# over the host threads (load_and_compile_s), the whole GPU suite at that library (the 8-shard files it writes
# carry the inference runs' empty bracket under its own name) and the smoke run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_PROFILE_STAGE=bench NSK_PROFILE_PARTIAL=1 NSK_PROFILE_SKIP_DEFAULT=1 NSK_PROFILE_BENCH_WORKLOADS="ising10m" \
  NSK_PROFILE_BENCH_ONLY="lr50m lr50m_learn" NSK_PROFILE_FULL_TESTS=1 bash tools/collect_profiles.sh
OUT=$R/gpurun_out/profiles_r5
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r5_lr50m*bench.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.4e" % d["value"], d.get("ms_per_step"), {k: d["config"].get(k) for k in ("generate_s", "load_and_compile_s", "compile_s")})
for f in sorted(glob.glob("$OUT/config5_shards_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d.get("exchange_fraction"), {k: round(v["mean"], 1) for k, v in d.get("per_shard_us", {}).items()})
PY
grep -E "positions|balancing|classes" $OUT/r5_lr50m_learn_compile_laps.txt
tail -3 $OUT/r5_gpu_tests.log
