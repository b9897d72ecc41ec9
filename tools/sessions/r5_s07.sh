#!/bin/bash
# Round 5, GPU session 7 (runs ON THE GPU BOX): the single-GPU kernels against the round-4 library ON ONE BOX
# (variants/libnsk_R4.so, built from commit 9f29698), the fused exchange whose interior runs take the single-GPU
# body, and the fused protocol's self-test.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s07; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/sweep  %.2f us/launch' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['avg_launch_us']))"; }
for REP in 1 2; do
for V in new R4; do
  if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
  for WL in ising10m ising10m_learn lr5m; do
    for ST in 20 200; do
      [ $WL != ising10m ] && [ $ST = 20 ] && continue
      echo -n "$WL $V steps $ST: " >> $OUT/bench.txt
      python bench.py --workload $WL --steps $ST --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
    done
  done
done
done
unset NSK_LIB
cat $OUT/bench.txt
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_config5_shards_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "not lr50m and not lr5m" > $OUT/shards.log 2>&1
echo "tests rc $? $(tail -1 $OUT/shards.log)"
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
