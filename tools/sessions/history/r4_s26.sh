#!/bin/bash
# round 4, session 26: profile stage of the partial re-collection at the round's last library (the workloads
# whose kernels or layout changed since the first collection), plus two quick A/Bs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
for r in 1 2; do
  echo -n "lr5m_learn (10 per CU): "; one lr5m_learn
  echo -n "lr5m_learn 7 per CU: "; NSK_DIAG=1 NSK_EP_PER_CU=7 one lr5m_learn
done
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m_learn 16 parts: "; NSK_DIAG=1 NSK_SHAPE_PARTS=16 one boolw4m_learn
echo -n "boolw4m_learn 32 parts: "; NSK_DIAG=1 NSK_SHAPE_PARTS=32 one boolw4m_learn
export NSK_PROFILE_PARTIAL=1 NSK_PROFILE_STAGE=profile NSK_PROFILE_WORKLOADS="lr5m_learn boolw4m boolw4m_learn" NSK_PROFILE_LIGHT_WORKLOADS="lr50m_learn"
bash tools/collect_profiles.sh
