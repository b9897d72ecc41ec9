#!/bin/bash
# round 4, session 8: after the always-zero id for ignored member slots -- LR 2-rank learning, the multi-rank tests, the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for lag in 0 1; do LAG=$lag timeout 300 python tools/debug/lr2.py 2>&1 | grep "^sweep" | grep -vc "diffs 0 \[\] evid diffs 0 \[\] weight diffs 0"; done
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q > gpurun_out/s08_multirank.log 2>&1; echo "multirank rc $?"; tail -3 gpurun_out/s08_multirank.log
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/s08_all.log 2>&1; echo "all rc $?"; tail -5 gpurun_out/s08_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/s08_bench.json 2> gpurun_out/s08_bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/s08_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['bound'], d['roofline']['time_over_memory_floor'], d['parity']['ok'], {k: v['value'] for k, v in d['also'].items()}, d['cpu_baseline']['value'], d['config']['compile_s'])"
