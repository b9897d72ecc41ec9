#!/bin/bash
# Round 5, GPU session 5 (runs ON THE GPU BOX): which half of the buffer addressing costs k_gibbs_seg_tab its
# 2.4 us (variants with flat loads / flat stores / both = the round-4 addressing); the quad scheme of all-evidence
# learning segments; border ordering of the fused exchange.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s05; rm -rf $OUT; mkdir -p $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for V in new FLATLD FLATST FLATBOTH; do
  if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
  for WL in ising10m ising1m; do
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_LIB
echo -n "ising10m_learn new " >> $OUT/bench.txt
python bench.py --workload ising10m_learn --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2> $OUT/learn.err | line >> $OUT/bench.txt
cat $OUT/bench.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_config5_shards_gpu.py tests/test_config4_gpu.py -m gpu -x -q -k "not lr50m and not lr5m" > $OUT/shards.log 2>&1
echo "shards rc $? $(tail -1 $OUT/shards.log)"
cp gpurun_out/config5_shards_*.json $OUT/ 2>/dev/null
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/two_ranks_one_device_bench.json 2> $OUT/two_ranks.err
python - <<PY
import json
d = json.loads(open("$OUT/two_ranks_one_device_bench.json").read().strip().splitlines()[-1])
print("two ranks: %.4e  %.2f us/sweep  launches %d  avg %.2f us  phases %s" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["launches"], d["roofline"]["avg_launch_us"], d.get("phases_us")))
PY
