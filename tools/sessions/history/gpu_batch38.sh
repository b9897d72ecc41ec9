#!/bin/bash
# the XCD self-test of nsk_graph_create: accumulator tests, learning parity, smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cabi.py tests/test_multirank_gpu.py -m gpu -x -q -k "accumulator or tiny_resident or learn or cabi or info or multirank" > gpurun_out/b38.log 2>&1; echo "rc $?"; tail -3 gpurun_out/b38.log
NSK_VERBOSE=1 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "self-test\|smoke" | tail -3
