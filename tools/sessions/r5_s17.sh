#!/bin/bash
# Round 5, GPU session 17 (runs ON THE GPU BOX): the closing partial collection at the round's last library --
# the LR bench lines (compile laps with the word cache; the learning sweep without the empty program-weight
# refresh), the two-rank LR learning line, the whole GPU suite (8-shard runs included) and the smoke run.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_PROFILE_STAGE=bench NSK_PROFILE_PARTIAL=1 NSK_PROFILE_SKIP_DEFAULT=1 NSK_PROFILE_BENCH_WORKLOADS="lr5m lr5m_learn" \
  NSK_PROFILE_BENCH_ONLY="lr50m lr50m_learn" NSK_PROFILE_FULL_TESTS=1 bash tools/collect_profiles.sh
OUT=$R/gpurun_out/profiles_r5
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload lr5m_learn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r5_two_ranks_one_device_lr5m_learn_bench.json 2>/dev/null
echo "two ranks lr5m_learn rc $?"
python bench.py --steps 20 --warmup 5 > $OUT/r5_driver_flags_bench.json 2> /dev/null
echo "driver-flags bench rc $?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r5_*bench.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.4e" % d["value"], d.get("ms_per_step"), d["config"].get("compile_s"), d["config"].get("generate_s"))
    except Exception as e:
        print(f, "unreadable", e)
for f in sorted(glob.glob("$OUT/config5_shards_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d.get("exchange_fraction"), {k: round(v["mean"], 1) for k, v in d.get("per_shard_us", {}).items()})
PY
tail -3 $OUT/r5_gpu_tests.log
