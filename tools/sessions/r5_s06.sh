#!/bin/bash
# Round 5, GPU session 6 (runs ON THE GPU BOX): state check after the reverts -- the single-GPU table kernel is the
# round-4 body again, border runs in front in the fused launches, relaxed flag polls and a 64-block weight gather in
# the exchange kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s06; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/default_bench.json 2> $OUT/default_bench.err
echo "default bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/default_bench.json").read().strip().splitlines()[-1])
print("ising10m %.4e  %.2f us/sweep  %.2f us/launch  stats %s" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_us"], d["parity"]["statistics"]))
for k, v in d["also"].items(): print(k, "%.4e  %.2f us/launch" % (v["value"], v["avg_launch_us"]))
PY
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_config5_shards_gpu.py -m gpu -x -q -k "not lr50m" > $OUT/shards.log 2>&1
echo "shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_*.json $OUT/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/config5_shards_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d.get("exchange_fraction"), {k: round(v["mean"], 1) for k, v in d["per_shard_us"].items()})
PY
for MODE in fused unfused; do
  if [ $MODE = unfused ]; then export NSK_DIAG=1 NSK_NO_P2P_FUSE=1; else unset NSK_DIAG NSK_NO_P2P_FUSE; fi
  NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/two_ranks_$MODE.json 2> $OUT/two_ranks_$MODE.err
  python - <<PY
import json
d = json.loads(open("$OUT/two_ranks_$MODE.json").read().strip().splitlines()[-1])
print("two ranks $MODE: %.4e  %.2f us/sweep  launches %d  avg %.2f us" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["launches"], d["roofline"]["avg_launch_us"]))
PY
done
unset NSK_DIAG NSK_NO_P2P_FUSE
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload lr5m_learn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/two_ranks_lr5m_learn.json 2> $OUT/two_ranks_lr5m_learn.err
python - <<PY
import json
d = json.loads(open("$OUT/two_ranks_lr5m_learn.json").read().strip().splitlines()[-1])
print("two ranks lr5m_learn: %.4e  %.1f us/sweep  phases %s" % (d["value"], d["ms_per_step"] * 1e3, d.get("phases_us")))
PY
