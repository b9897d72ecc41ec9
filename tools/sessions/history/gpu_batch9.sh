#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x > gpurun_out/b9_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b9_pytest.log
run() { NSK_LIB=$2 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.1f us/launch  launches %d  layoutB %.1f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['layout_bytes_per_update']))"; }
echo -n "lr5m base: "; run lr5m ""
echo -n "lr5m_learn base: "; run lr5m_learn ""
echo -n "lr5m gen_block=1024: "; NSK_DIAG=1 NSK_GEN_BLOCK=1024 run lr5m ""
echo -n "lr5m_learn gen_block=1024: "; NSK_DIAG=1 NSK_GEN_BLOCK=1024 run lr5m_learn ""
bash tools/trace_dispatches.sh lr5m_learn 3 2>&1 | tail -14
