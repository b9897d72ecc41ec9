#!/bin/bash
# round 4, session 18: the whole GPU suite, smoke and the default bench at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q > gpurun_out/s18_all.log 2>&1; echo "all rc $?"; tail -4 gpurun_out/s18_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/s18_bench.json 2> gpurun_out/s18_bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/s18_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['bound'], d['roofline']['time_over_memory_floor'], d['parity']['ok'], {k: v['value'] for k, v in d['also'].items()}, d['cpu_baseline']['value'], d['config']['compile_s'])"
