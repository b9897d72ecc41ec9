#!/bin/bash
# Round 5, GPU session 23 (runs ON THE GPU BOX): the table kernels with the tally bytes (k_gibbs_seg_tab) and the
# evidence values (k_learn_seg_tab) requested BEHIND the member gathers instead of in front of them -- in front, the
# compiler's vmcnt(0) for the stream words of non-affine segments made an affine tile wait for them before its
# gathers were issued: a round trip per trip.  Parity of the grids, then the default line and the grid lines
# against the library of the commit before (libnsk_BASE.so: also without the entry-parallel change) on this box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r5_s23; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q > $OUT/parity.log 2>&1
echo "parity rc $? $(tail -1 $OUT/parity.log)"
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_config4_gpu.py -m gpu -x -q -k "grid or config4" > $OUT/ranks.log 2>&1
echo "ranks rc $? $(tail -1 $OUT/ranks.log)"
for V in new BASE new BASE; do
  if [ $V = new ]; then unset NSK_LIB; else export NSK_LIB=$R/numbskull_amd/variants/libnsk_$V.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/default_$V.json 2> $OUT/default_$V.err
  python - <<PY
import json
d = json.loads(open("$OUT/default_$V.json").read().strip().splitlines()[-1])
print("$V ising10m %.4e  %.2f us/sweep  %.2f us/launch" % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_us"]), {k: ("%.4e" % v["value"], round(v["avg_launch_us"], 2)) for k, v in d["also"].items()})
PY
done
unset NSK_LIB
