#!/bin/bash
# round-3 batch 1 (runs on the GPU box): parity suite, default bench, LR block-size A/B, LR PMC bytes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_config5_gpu.py --durations=12 > gpurun_out/b1_pytest.log 2>&1
echo "pytest rc $?"; tail -25 gpurun_out/b1_pytest.log
python bench.py > gpurun_out/b1_bench_default.json 2> gpurun_out/b1_bench_default.err
echo "bench rc $?"; cut -c1-600 gpurun_out/b1_bench_default.json
for gb in default 2048 8192 32768; do
  if [ $gb = default ]; then python bench.py --workload lr5m --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b1_lr5m_$gb.json 2>/dev/null
  else NSK_DIAG=1 NSK_GEN_BLOCK=$gb python bench.py --workload lr5m --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b1_lr5m_$gb.json 2>/dev/null; fi
  echo "lr5m gen_block=$gb: $(python - <<PY
import json
d=json.loads(open("gpurun_out/b1_lr5m_$gb.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["layout_bytes_per_update"], d["config"]["colors"])
PY
)"
done
tools/pmc.sh lr5m_fetch FETCH_SIZE --workload lr5m
tools/pmc.sh lr5m_write WRITE_SIZE --workload lr5m
tools/pmc.sh lr5ml_fetch FETCH_SIZE --workload lr5m_learn
tools/pmc.sh lr5ml_write WRITE_SIZE --workload lr5m_learn
find gpurun_out -type f -size +4M -delete
