#!/bin/bash
# round 4, session 9: entry-parallel kernels with capped vector registers (more resident waves); 50M compile laps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-50} --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch  compile %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s']))"; }
for v in new G5 G6; do lib=""; [ $v != new ] && lib=$R/numbskull_amd/variants/libnsk_$v.so; echo -n "lr5m $v: "; NSK_LIB=$lib one lr5m; done
for v in new L4 L5; do lib=""; [ $v != new ] && lib=$R/numbskull_amd/variants/libnsk_$v.so; echo -n "lr5m_learn $v: "; NSK_LIB=$lib one lr5m_learn; done
for v in G5 G6; do echo -n "boolw4m $v: "; NSK_LIB=$R/numbskull_amd/variants/libnsk_$v.so one boolw4m; done
echo -n "boolw4m new: "; one boolw4m; echo -n "boolw4m_learn new: "; one boolw4m_learn
NSK_VERBOSE=1 python bench.py --workload lr50m --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2> gpurun_out/s09_lr50m.err | tail -1 > gpurun_out/s09_lr50m.json
python -c "import json; d=json.load(open('gpurun_out/s09_lr50m.json')); print('lr50m %.4e updates/s %.2f us/launch compile %.1f s load %.1f s gen %.1f s' % (d['value'], d['roofline']['avg_launch_us'], d['config']['compile_s'], d['config']['load_and_compile_s'], d['config']['generate_s']))"
grep "compile " gpurun_out/s09_lr50m.err | head -30 | tee gpurun_out/s09_lr50m_laps.txt
for v in G5 G6; do echo -n "lr50m $v: "; NSK_LIB=$R/numbskull_amd/variants/libnsk_$v.so one lr50m 10; done
