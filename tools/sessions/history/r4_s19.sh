#!/bin/bash
# round 4, session 19: entry-parallel launch parameters at 50M (workgroups per CU, id-block size of the group order)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
one() { python bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"; }
echo -n "lr50m default: "; one lr50m
for pcu in 4 10; do echo -n "lr50m EP_PER_CU $pcu: "; NSK_DIAG=1 NSK_EP_PER_CU=$pcu one lr50m; done
for blk in 256 4096; do echo -n "lr50m EP_BLOCK $blk: "; NSK_DIAG=1 NSK_EP_BLOCK=$blk one lr50m; done
echo -n "lr50m_learn default: "; one lr50m_learn
for pcu in 4 10; do echo -n "lr50m_learn EP_PER_CU $pcu: "; NSK_DIAG=1 NSK_EP_PER_CU=$pcu one lr50m_learn; done
