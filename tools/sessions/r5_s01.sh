#!/bin/bash
# Round 5, GPU session 1 (runs ON THE GPU BOX):
#  (a) the default bench line at the driver's flags (--steps 20 --warmup 5) with the closing event recorded but
#      not waited for inside the timed region, and >= 10 untimed burn-in sweeps in front;
#  (b) where k_learn_ep's bytes go: variants of the learning translation unit with one stage
#      removed (tools/build_ablations.sh, NSK_ABL_TU=learn), kernel time and FETCH_SIZE / WRITE_SIZE per launch on
#      the 5M and 50M LR graphs.  The variants compute wrong samples by construction; they price the stages.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s01; rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/default_bench.json 2> $OUT/default_bench.err
echo "default bench rc $?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "TCC_EA0_R[A-Z0-9_]*\|TCC_EA0_W[A-Z0-9_]*\|TCC_ATOMIC[A-Z0-9_]*\|TCC_EA0_ATOMIC[A-Z0-9_]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/avail_tcc.txt
summarise() {   # dir tag
python3 - "$1" "$2" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
for fn in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "learn_ep" in r["Name"] or "apply_weights" in r["Name"] or "gibbs_ep" in r["Name"]:
            print("%s time %-40s calls %s avg_us %.1f" % (tag, r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
res = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        res[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in res.items():
    if "learn_ep" in k or "gibbs_ep" in k:
        for cn, v in sorted(c.items()):
            print("%s pmc %-40s %-22s mean %.5g n %d" % (tag, k, cn, sum(v) / len(v), len(v)))
PY
}
for WL in lr5m_learn lr50m_learn; do
  for V in new NOATOMIC EPNOW EPNOVAL EPNOP3 EPNOW+EPNOVAL+NOATOMIC; do
    [ $WL = lr50m_learn ] && [ $V = EPNOP3 ] && continue
    if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
    D=$OUT/${WL}_$V; mkdir -p $D
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $D/trace.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $D/fetch.log 2>&1
    if [ $WL = lr5m_learn ] || [ $V = new ]; then
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $D/write.log 2>&1
    fi
    summarise $D "$WL $V" >> $OUT/summary.txt
    find $D -type f -size +1M -delete
  done
done
unset NSK_LIB
cat $OUT/summary.txt
tail -c 600 $OUT/default_bench.json
