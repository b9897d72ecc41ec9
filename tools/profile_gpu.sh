#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench plus the
# PMC passes for HBM traffic, each counter set in its own run (MI355X_MICROARCH.md, HBM section).
# Raw output goes to gpurun_out/prof/; tools/summarize_profile.py condenses it into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-ising10m}
OUT=$R/gpurun_out/prof_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
WL=${1:-ising10m}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --steps ${NSK_PROFILE_STEPS:-100} --warmup 10 --no-cpu-baseline --no-extra > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_write.log 2>&1
if [ -z "$NSK_PROFILE_LIGHT" ]; then      # (light: kernel stats + HBM traffic only -- the 50M graphs take minutes per run)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_l2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_sq.log 2>&1
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/calib_stream.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -- python3 $R/tools/calib_stream.py > $OUT/calib_write.log 2>&1
python3 $R/tools/summarize_profile.py $OUT $WL > $OUT/summary.txt 2>&1
# keep the merge small: drop anything big
find $OUT -type f -size +4M -delete

cat $OUT/summary.txt
