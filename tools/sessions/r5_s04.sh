#!/bin/bash
# Round 5, GPU session 4 (runs ON THE GPU BOX): k_gibbs_seg_tab with buffer addressing and every load of a quad
# in flight again; the fused boundary exchange with system-coherent loads / stores instead of fences.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s04; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_config5_shards_gpu.py -m gpu -x -q -k "not lr50m and not lr5m" > $OUT/shards.log 2>&1
echo "shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_*.json $OUT/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/default_bench.json 2> $OUT/default_bench.err
echo "default bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/default_bench.json").read().strip().splitlines()[-1])
print("ising10m %.4e  %.2f us/sweep  %.2f us/launch" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_us"]))
for k, v in d["also"].items(): print(k, "%.4e  %.2f us/launch" % (v["value"], v["avg_launch_us"]))
PY
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in ising10m lr5m_learn; do
  echo -n "$WL " >> $OUT/bench.txt
  python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/${WL}.err | line >> $OUT/bench.txt
done
cat $OUT/bench.txt
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/two_ranks_one_device_bench.json 2> $OUT/two_ranks.err
echo "two ranks rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/two_ranks_one_device_bench.json").read().strip().splitlines()[-1])
print("two ranks: %.4e  %.2f us/sweep  launches %d  avg %.2f us  phases %s" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["launches"], d["roofline"]["avg_launch_us"], d.get("phases_us")))
PY
NSK_DIAG=1 NSK_NO_P2P_FUSE=1 NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/two_ranks_unfused.json 2> $OUT/two_ranks_unfused.err
python - <<PY
import json
d = json.loads(open("$OUT/two_ranks_unfused.json").read().strip().splitlines()[-1])
print("two ranks, exchange kernels: %.4e  %.2f us/sweep  launches %d  avg %.2f us" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["launches"], d["roofline"]["avg_launch_us"]))
PY
