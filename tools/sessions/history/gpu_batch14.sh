#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_hip_parity.py -m gpu -q -x -k "two_ranks or partition or rccl or shard" > gpurun_out/b14_pytest.log 2>&1
echo "pytest rc $?"; tail -6 gpurun_out/b14_pytest.log
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/b14_bench2.err | tail -1 | cut -c1-1500
tail -5 gpurun_out/b14_bench2.err
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload lr5m_learn --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/b14_bench2l.err | tail -1 | cut -c1-800
tail -5 gpurun_out/b14_bench2l.err
