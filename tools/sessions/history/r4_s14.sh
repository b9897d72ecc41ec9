#!/bin/bash
# round 4, session 14: one-launch peer-to-peer exchange for the inference loops; downloads into the caller's arrays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/profiles_r4
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "two_ranks or peer or captured or cabi or state or continue" > gpurun_out/s14_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/s14_tests.log
timeout 900 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "grid10m" > gpurun_out/s14_shards.log 2>&1; echo "shards rc $?"; tail -2 gpurun_out/s14_shards.log
python tools/debug/xfer_time.py 2>&1 | tail -4
for i in 1 2; do
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/profiles_r4/r4_two_ranks_one_device_bench.json
python -c "import json; d=json.load(open('gpurun_out/profiles_r4/r4_two_ranks_one_device_bench.json')); print('2 ranks ising10m', d['value'], d['ms_per_step'], d.get('phases_us'), d['parity']['ok'])"
done
python bench.py --workload lr5m --steps 100 --warmup 10 --no-extra > gpurun_out/profiles_r4/r4_lr5m_bench.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/profiles_r4/r4_lr5m_bench.json')); print('lr5m', d['value'], d['roofline']['avg_launch_us'], d['roofline']['kernel'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
