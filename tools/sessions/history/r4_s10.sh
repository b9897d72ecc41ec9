#!/bin/bash
# round 4, session 10: table kernel with the closing round dealt as single tiles (A/B against HEAD~), parity of the table paths
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q -k "table or grid or captured or domain or shards_draw or tiny or exact or config" > gpurun_out/s10_parity.log 2>&1; echo "parity rc $?"; tail -3 gpurun_out/s10_parity.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
NSK_VARIANTS="new OLD" bash tools/ab_lib.sh "ising10m ising40m ising1m" 200
NSK_VARIANTS="new OLD" bash tools/ab_lib.sh "ising10m ising40m" 200
