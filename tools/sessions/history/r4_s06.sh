#!/bin/bash
# round 4, session 6: table-path learning mismatch with the sat-bit check build; compile laps and bench lines of the LR graphs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
env LAG=0 NSK_LIB=$R/numbskull_amd/variants/libnsk_LCHECK.so timeout 300 python tools/debug/lr2.py > gpurun_out/s06_lcheck.log 2>&1
grep -c LCHECK gpurun_out/s06_lcheck.log; grep "MISMATCH" gpurun_out/s06_lcheck.log | head -5; grep "LCHECK" gpurun_out/s06_lcheck.log | head -12; grep "^sweep" gpurun_out/s06_lcheck.log | head -4 | cut -c1-250
for w in lr5m lr5m_learn; do
  python bench.py --workload $w --steps 50 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/s06_$w.json
  python -c "import json; d=json.load(open('gpurun_out/s06_$w.json')); print('$w %.4e updates/s %.2f us/launch compile %.1f s load %.1f s gen %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s'], d['config']['load_and_compile_s'], d['config']['generate_s']))"
done
for w in lr50m lr50m_learn; do
  NSK_VERBOSE=1 python bench.py --workload $w --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2> gpurun_out/s06_$w.err | tail -1 > gpurun_out/s06_$w.json
  python -c "import json; d=json.load(open('gpurun_out/s06_$w.json')); print('$w %.4e updates/s %.2f us/launch compile %.1f s load %.1f s gen %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s'], d['config']['load_and_compile_s'], d['config']['generate_s']))"
  grep "compile " gpurun_out/s06_$w.err | head -30 > gpurun_out/s06_${w}_laps.txt
done
cat gpurun_out/s06_lr50m_laps.txt
