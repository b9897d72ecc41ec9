#!/bin/bash
# round 4, session 11: table kernel with 94 scalar registers (7 blocks per CU resident); tally download; two-rank bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for w in ising10m ising40m; do
  echo -n "$w new: "; one $w
  for cap in -1536 -1792; do echo -n "$w S96 cap $cap: "; NSK_LIB=$R/numbskull_amd/variants/libnsk_S96.so NSK_DIAG=1 NSK_TAB_GRID_CAP=$cap one $w; done
done
python tools/debug/xfer_time.py 2>&1 | tail -4
timeout 300 python -m pytest tests/test_cabi.py tests/test_hip_parity.py -m gpu -x -q -k "cabi or state or upload or continue or domain or config1 or cli" 2>&1 | tail -2
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/s11_two.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks ising10m', d['value'], d['ms_per_step'], d.get('phases_us'), d['parity']['ok'])"
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload lr5m_learn --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/s11_two_lr.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks lr5m_learn', d['value'], d['ms_per_step'], d.get('phases_us'), d['parity']['ok'], d['config']['generate_s'], d['config']['partition'])"
tail -3 gpurun_out/s11_two_lr.err
