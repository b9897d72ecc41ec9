#!/bin/bash
# round 4, session 5: localise the table-path learning mismatch; table kernel with batched loads and a pair-dealt tail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for sw in REMAP NSK_NO_AFFINE; do
  echo "== $sw"
  env LAG=0 NSK_DIAG=1 $sw=1 timeout 300 python tools/debug/lr2.py > gpurun_out/s05_lr2_$sw.log 2>&1
  grep "^remapped\|^sweep" gpurun_out/s05_lr2_$sw.log | grep -v "weight diffs 0" | head -4 | cut -c1-400
done
echo "== grid cap 8"; env LAG=0 NSK_DIAG=1 NSK_LEARN_GRID_CAP=8 timeout 300 python tools/debug/lr2.py 2>&1 | grep "^sweep" | grep -v "weight diffs 0" | head -3 | cut -c1-300
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "table or grid or captured or domain or shards_draw or tiny or exact or config" > gpurun_out/s05_parity.log 2>&1; echo "parity rc $?"; tail -3 gpurun_out/s05_parity.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
NSK_VARIANTS="new TB2 OLD" bash tools/ab_lib.sh "ising10m ising40m ising1m" 200
