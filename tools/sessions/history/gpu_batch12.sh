#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --workload $1 --steps ${2:-20} --warmup 3 --no-cpu-baseline 2>gpurun_out/b12_$1.err | tail -1 > gpurun_out/b12_$1.json; python -c "import json,sys; d=json.loads(open('gpurun_out/b12_$1.json').read()); print('%.3e updates/s  %.1f us/launch  launches %d  layoutB %.1f colors %d compile %.1fs' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['layout_bytes_per_update'], d['config']['colors'], d['config']['compile_s']))"; }
echo -n "lr5m: "; run lr5m
echo -n "lr5m_learn: "; run lr5m_learn
echo -n "lr50m: "; run lr50m 10
echo -n "lr50m_learn: "; run lr50m_learn 10
echo -n "boolw4m: "; run boolw4m
echo -n "boolw4m_learn: "; run boolw4m_learn
