#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_multirank_gpu.py tests/test_config3_gpu.py tests/test_config4_gpu.py -m gpu -x -q > gpurun_out/b26_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b26_pytest.log
for wl in ising10m ising1m ising64k; do for ng in 0 1; do
echo -n "$wl NO_GRAPH=$ng: "; NSK_DIAG=$ng NSK_NO_GRAPH=$ng python bench.py --workload $wl --steps 400 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch ms/step %.4f' % (d['value'], d['roofline']['avg_launch_us'], d['ms_per_step']))"
done; done
for p in 1 0; do
echo -n "2 ranks one device NSK_P2P=$p: "; NSK_P2P=$p NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e ms/step %.4f' % (d['value'], d['ms_per_step']), d.get('phases_us'))"
done
