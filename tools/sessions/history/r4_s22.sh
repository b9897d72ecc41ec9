#!/bin/bash
# round 4, session 22: where the learning sweep of the weighted boolean graph spends its time (shape tiles
# walked per lane vs entry-parallel groups), its profile, and one accumulator copy at 50M
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn NO_SHAPE: "; NSK_DIAG=1 NSK_NO_SHAPE=1 one boolw4m_learn
echo -n "boolw4m_learn NO_EP: "; NSK_DIAG=1 NSK_NO_EP=1 one boolw4m_learn
echo -n "boolw4m_learn NO_SHAPE NO_EP: "; NSK_DIAG=1 NSK_NO_SHAPE=1 NSK_NO_EP=1 one boolw4m_learn
echo -n "boolw4m NO_SHAPE: "; NSK_DIAG=1 NSK_NO_SHAPE=1 one boolw4m
for pcu in 6 8; do echo -n "boolw4m_learn EP_PER_CU $pcu: "; NSK_DIAG=1 NSK_EP_PER_CU=$pcu one boolw4m_learn; done
bash tools/profile_gpu.sh boolw4m_learn > gpurun_out/s22_boolw4m_learn_summary.txt 2>&1
grep -v "seg_tab\|calibration" gpurun_out/s22_boolw4m_learn_summary.txt | head -40
echo -n "lr50m_learn: "; one lr50m_learn 10
echo -n "lr50m_learn ONE_ACC: "; NSK_DIAG=1 NSK_ONE_ACC=1 one lr50m_learn 10
