#!/bin/bash
# Round 6, GPU session 1 (runs ON THE GPU BOX): first run of the wide-quad table kernel -- smoke, the parity tests of
# the table kernels, and the headline / 1M benches at the driver's flags.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/r6_s01
O=gpurun_out/r6_s01
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
for WL in ising10m ising1m ising40m; do
  timeout 600 python bench.py --steps 20 --warmup 5 --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err; echo "bench $WL rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$WL.json").read().strip().splitlines()[-1])
    print("$WL", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("parity"))
except Exception as e:
    print("parse failed", e)
PY
done
