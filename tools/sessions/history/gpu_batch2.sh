#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_multirank_gpu.py -m gpu -q > gpurun_out/b2_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/b2_pytest.log
for wl in lr5m lr5m_learn; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b2_$wl.json 2>gpurun_out/b2_$wl.err
  echo "$wl: $(python - <<PY
import json
d=json.loads(open("gpurun_out/b2_$wl.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["layout_bytes_per_update"], d["config"]["colors"], d.get("learn_clipped"))
PY
)"
done
true

