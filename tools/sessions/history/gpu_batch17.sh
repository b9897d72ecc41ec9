#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py tests/test_learning_tie_gpu.py -m gpu -x -q --deselect tests/test_learning_tie_gpu.py::test_many_weight_lr_graph_chromatic_matches_sequential > gpurun_out/b17_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b17_pytest.log
for wl in ising10m_learn ising1m_learn ising1m ising10m; do
echo -n "$wl: "; python bench.py --workload $wl --steps 50 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch frac %.3f launches %d clipped %s' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['launches'], d.get('learn_clipped')))"
done
bash tools/trace_dispatches.sh ising10m_learn 3 2>&1 | tail -8
