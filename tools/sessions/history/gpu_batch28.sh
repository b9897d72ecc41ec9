#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_config4_gpu.py -m gpu -x -q 2>&1 | tail -4
for p in 1 0; do
echo -n "2 ranks one device NSK_P2P=$p: "; NSK_P2P=$p NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline 2>gpurun_out/b28_$p.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e ms/step %.4f' % (d['value'], d['ms_per_step']), d.get('phases_us'), d['parity'].get('ok'))"
tail -3 gpurun_out/b28_$p.err | grep -v socket
done
echo -n "2 ranks p2p NO_GRAPH: "; NSK_DIAG=1 NSK_NO_GRAPH=1 NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e ms/step %.4f' % (d['value'], d['ms_per_step']))"
