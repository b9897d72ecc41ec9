#!/bin/bash
# round 4, session 13: the register-capped twin of the entry-parallel inference kernel, captured peer-to-peer
# sequences up to 6M variables, state transfer with cacheable staging; lr5m re-profiled
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/s13_parity.log 2>&1; echo "parity rc $?"; tail -3 gpurun_out/s13_parity.log
timeout 900 python -m pytest tests/test_config5_shards_gpu.py -m gpu -x -q -k "lr5m" > gpurun_out/s13_shards.log 2>&1; echo "shards rc $?"; tail -2 gpurun_out/s13_shards.log
python tools/debug/xfer_time.py 2>&1 | tail -4
one() { python bench.py --workload $1 --steps ${2:-100} --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "lr5m: "; one lr5m; echo -n "lr50m: "; one lr50m 10
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/s13_two_ranks.json
python -c "import json; d=json.load(open('gpurun_out/s13_two_ranks.json')); print('2 ranks ising10m', d['value'], d['ms_per_step'], d.get('phases_us'), d['parity']['ok'])"
NSK_PROFILE_STAGE=profile NSK_PROFILE_WORKLOADS=lr5m NSK_PROFILE_LIGHT_WORKLOADS="" NSK_PROFILE_PARTIAL=1 bash tools/collect_profiles.sh | tail -3
