#!/bin/bash
# round 4, session 27: profile stage of the partial re-collection at the round's last library (16 id ranges,
# rest-tile workgroups, 7 / 10 workgroups per CU by group count), preceded by the quick parity check of the
# kernel-level tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x 2>&1 | tail -2
one() { python bench.py --workload $1 --steps ${2:-30} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "boolw4m_learn: "; one boolw4m_learn
echo -n "boolw4m: "; one boolw4m
echo -n "lr5m_learn: "; one lr5m_learn
export NSK_PROFILE_PARTIAL=1 NSK_PROFILE_STAGE=profile NSK_PROFILE_WORKLOADS="lr5m_learn boolw4m boolw4m_learn" NSK_PROFILE_LIGHT_WORKLOADS="lr50m_learn"
bash tools/collect_profiles.sh
