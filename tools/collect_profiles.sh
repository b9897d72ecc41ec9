#!/bin/bash
# Runs ON THE GPU BOX (gpurun): regenerates EVERY profiles/${RT}_* artefact at the current HEAD in one go --
# the default bench line, kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE, L2, SQ; each counter
# set in its own run, calibrated on a known-byte stream copy of the same run) for the workloads listed,
# the bench lines of the large configurations, and the HBM-traffic table bench.py reads.  Output:
# gpurun_out/profiles_$RT/ (tools/install_profiles.sh copies it into profiles/).
# NSK_PROFILE_PARTIAL=1 with NSK_PROFILE_WORKLOADS / NSK_PROFILE_BENCH_ONLY: only those workloads (a change
# that touches some kernel families only); the traffic table keeps the other workloads' entries and
# install_profiles.sh replaces only the files collected.
# NSK_PROFILE_STAGE=profile: the rocprofv3 passes and the two tables only; =bench: the bench lines, the two-rank
# lines and the 8-shard runs only (they read profiles/traffic.json / issue.json as installed from a profile
# stage: a gpurun call is limited to an hour and the whole collection takes longer); default: both.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RT=${NSK_ROUND_TAG:-r6}
STAGE=${NSK_PROFILE_STAGE:-all}
OUT=$R/gpurun_out/profiles_$RT
rm -rf $OUT; mkdir -p $OUT
TRAFFIC=$OUT/traffic_parts; mkdir -p $TRAFFIC
if [ $STAGE != bench ]; then
for WL in ${NSK_PROFILE_WORKLOADS:-ising10m ising10m_learn ising1m lr5m lr5m_learn boolw4m boolw4m_learn}; do
  bash tools/profile_gpu.sh $WL > /dev/null 2>&1
  P=$R/gpurun_out/prof_$WL
  cp $P/summary.txt $OUT/${RT}_${WL}_summary.txt 2>/dev/null
  cp $P/summary.json $OUT/${RT}_${WL}_summary.json 2>/dev/null
  f=$(find $P/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${RT}_${WL}_kernel_stats.csv
  cp $P/traffic_$WL.json $P/issue_$WL.json $TRAFFIC/ 2>/dev/null
  echo "profiled $WL: $(grep -h 'dominant kernel' $P/summary.txt)"
done
# config #5 at size: kernel stats and HBM traffic only (minutes per run)
for WL in ${NSK_PROFILE_LIGHT_WORKLOADS:-lr50m lr50m_learn ising100m}; do      # (ising100m: the one grid whose sweep is beyond the Infinity Cache)
  NSK_PROFILE_LIGHT=1 NSK_PROFILE_STEPS=10 bash tools/profile_gpu.sh $WL > /dev/null 2>&1
  P=$R/gpurun_out/prof_$WL
  cp $P/summary.txt $OUT/${RT}_${WL}_summary.txt 2>/dev/null
  f=$(find $P/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${RT}_${WL}_kernel_stats.csv
  cp $P/traffic_$WL.json $TRAFFIC/ 2>/dev/null
  echo "profiled (light) $WL: $(grep -h 'dominant kernel' $P/summary.txt)"
done
# the HBM-traffic table first (bench.py prints it as roofline.traffic), then every bench line with it
python - <<PY
import json, glob, os
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh (tools/profile_gpu.sh per workload), "
                  "corrected with the known-byte stream-copy calibration of the same run; all workloads collected in the same "
                  "gpurun call as the bench lines (commit: profiles/${RT}_COMMIT.txt); separate passes from the bench run"}
if os.environ.get("NSK_PROFILE_PARTIAL") and os.path.exists("$R/profiles/traffic.json"):
    prev = json.load(open("$R/profiles/traffic.json"))
    out.update({k: v for k, v in prev.items() if not k.startswith("_")})
for f in sorted(glob.glob("$TRAFFIC/traffic_*.json")):
    out.update(json.load(open(f)))
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
json.dump(out, open("$R/profiles/traffic.json", "w"), indent=1)
print(json.dumps(out)[:600])
iss = {"_source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD pass of "
                  "tools/profile_gpu.sh per workload: means per launch of the dominant kernel"}
if os.environ.get("NSK_PROFILE_PARTIAL") and os.path.exists("$R/profiles/issue.json"):
    prev = json.load(open("$R/profiles/issue.json"))
    iss.update({k: v for k, v in prev.items() if not k.startswith("_")})
for f in sorted(glob.glob("$TRAFFIC/issue_*.json")):
    iss.update(json.load(open(f)))
json.dump(iss, open("$OUT/issue.json", "w"), indent=1)
json.dump(iss, open("$R/profiles/issue.json", "w"), indent=1)
PY
fi
[ $STAGE = profile ] && { ls $OUT; exit 0; }
if [ -z "$NSK_PROFILE_SKIP_DEFAULT" ]; then      # (a partial collection of one kernel family skips the default and two-rank lines)
python bench.py > $OUT/${RT}_default_bench.json 2> $OUT/${RT}_default_bench.err
echo "default bench rc $?"
python bench.py --steps 20 --warmup 5 > $OUT/${RT}_driver_flags_bench.json 2> /dev/null       # (the flags of the driver's own run)
echo "driver-flags bench rc $?"
fi
for WL in ${NSK_PROFILE_BENCH_WORKLOADS:-ising10m ising10m_learn ising1m lr5m lr5m_learn boolw4m boolw4m_learn}; do
  [ $WL = ising10m ] && continue        # (the default line)
  python bench.py --workload $WL --steps 100 --warmup 10 --no-extra > $OUT/${RT}_${WL}_bench.json 2> /dev/null
  echo "bench $WL rc $?"
done
for WL in ${NSK_PROFILE_BENCH_ONLY:-ising40m ising100m lr50m lr50m_learn}; do
  NSK_VERBOSE=1 python bench.py --workload $WL --steps 10 --warmup 3 --no-extra > $OUT/${RT}_${WL}_bench.json 2> $OUT/${RT}_${WL}_bench.err
  rc=$?
  grep "compile " $OUT/${RT}_${WL}_bench.err > $OUT/${RT}_${WL}_compile_laps.txt
  echo "bench $WL rc $rc"
done
[ -z "$NSK_PROFILE_SKIP_DEFAULT" ] && NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${RT}_two_ranks_one_device_bench.json 2>/dev/null
# (the same two ranks through the exchange kernels instead of the fused class launches, for comparison)
[ -z "$NSK_PROFILE_SKIP_DEFAULT" ] && NSK_DIAG=1 NSK_NO_P2P_FUSE=1 NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${RT}_two_ranks_one_device_exchange_kernels_bench.json 2>/dev/null
[ -z "$NSK_PROFILE_SKIP_DEFAULT" ] && NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload lr5m_learn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${RT}_two_ranks_one_device_lr5m_learn_bench.json 2>/dev/null
# the 8-shard runs on one device (per-shard phase timings): config #4 through pack / unpack, configs #4 and #5
# through the peer-to-peer kernels (the 50M graph included when the host has the memory)
rm -f gpurun_out/config4_shards_*.json gpurun_out/config5_shards_*.json
if [ -n "$NSK_PROFILE_FULL_TESTS" ]; then      # the whole GPU suite (it contains the 8-shard runs) in the same call
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/${RT}_gpu_tests.log 2>&1
  echo "gpu tests rc $? $(tail -1 $OUT/${RT}_gpu_tests.log)"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${RT}_smoke.log 2>&1
  echo "smoke rc $? $(tail -1 $OUT/${RT}_smoke.log)"
else
timeout 2400 python -m pytest tests/test_config5_shards_gpu.py tests/test_config4_gpu.py -m gpu -q > $OUT/${RT}_shards_tests.log 2>&1
echo "shard tests rc $? $(tail -1 $OUT/${RT}_shards_tests.log)"
fi
cp gpurun_out/config4_shards_*.json gpurun_out/config5_shards_*.json $OUT/ 2>/dev/null
find $OUT -type f -size +2M -delete
ls $OUT
