#!/bin/bash
# Round 5, GPU session 2 (runs ON THE GPU BOX): LDS value windows of the entry-parallel groups + non-temporal
# weight gathers (k_gibbs_ep / k_learn_ep), and the reduce-scatter / all-gather weight merge of the peer-to-peer
# learning exchange.
#  (a) parity: the chromatic parity tests, the 8-shard LR / grid runs, the two-process runs;
#  (b) timing + FETCH_SIZE of the LR workloads: this build, the windows switched off (NSK_NO_EP_WIN), temporal
#      weight gathers (WTEMP), non-temporal rows in learning (ROWNT).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s02; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chromatic or general or duplicate or exercised or ghost or partition or peer or accumulator or unpacked or edge_case or one_factor or shape" > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_config5_shards_gpu.py -m gpu -x -q -k "not lr50m" > $OUT/shards.log 2>&1
echo "shards rc $? $(tail -1 $OUT/shards.log)"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for WL in lr5m_learn lr5m lr50m_learn lr50m; do
  for V in new nowin WTEMP ROWNT; do
    case $V in
      new) unset NSK_LIB NSK_DIAG NSK_NO_EP_WIN;;
      nowin) unset NSK_LIB; export NSK_DIAG=1 NSK_NO_EP_WIN=1;;
      *) unset NSK_DIAG NSK_NO_EP_WIN; export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so;;
    esac
    case $WL in lr5m|lr50m) [ $V = WTEMP ] || [ $V = ROWNT ] && continue;; esac       # (learning-unit variants)
    [ $WL = lr50m ] && [ $V = nowin ] && continue
    echo -n "$WL $V " >> $OUT/bench.txt
    python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2> $OUT/${WL}_$V.err | line >> $OUT/bench.txt
  done
done
unset NSK_LIB NSK_DIAG NSK_NO_EP_WIN
cat $OUT/bench.txt
cd /tmp; export TMPDIR=/tmp
for WL in lr5m_learn lr50m_learn lr50m; do
  D=$OUT/fetch_$WL
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $D.log 2>&1
  python3 - $D $WL <<'PY'
import csv, glob, sys, collections
res = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "_ep<" in r["Kernel_Name"]: res[r["Kernel_Name"].split("(")[0][:44]].append(float(r["Counter_Value"]))
for k, v in res.items(): print(sys.argv[2], k, "FETCH_SIZE mean %.5g (x2 KB) n %d" % (sum(v) / len(v), len(v)))
PY
  find $D -type f -size +1M -delete
done
