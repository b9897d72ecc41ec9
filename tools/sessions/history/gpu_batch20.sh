#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "lr or gencat or hubs or general or dup" > gpurun_out/b20_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/b20_pytest.log
run() { NSK_LIB=$2 python bench.py --workload $1 --steps ${3:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
V=$R/numbskull_amd/variants
echo -n "lr5m base: "; run lr5m ""
echo -n "lr5m_learn base: "; run lr5m_learn ""
echo -n "lr50m base: "; run lr50m "" 5
echo -n "lr50m EPNOVAL: "; run lr50m $V/libnsk_EPNOVAL.so 5
