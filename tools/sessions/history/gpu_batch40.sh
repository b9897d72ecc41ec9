#!/bin/bash
# final check of the library as it leaves the round: config-#3 parity, tiny-grid tests, smoke, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_config3_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "config3 or tiny_resident or accumulator or captured" > gpurun_out/b40.log 2>&1; echo "rc $?"; tail -2 gpurun_out/b40.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['parity']['ok'], {k: v['value'] for k, v in d['also'].items()}, d['cpu_baseline']['value'])"
