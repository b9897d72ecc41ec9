#!/bin/bash
# Round 5, GPU session 35 (runs ON THE GPU BOX): the library as it is committed last -- smoke and the chromatic
# parity cases (the last rebuilds changed comments, diagnostics and verbose prints only).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT0=$R/gpurun_out/r5_s35; rm -rf $OUT0; mkdir -p $OUT0
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT0/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $OUT0/smoke.log)"
timeout 120 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "chromatic or general" > $OUT0/parity.log 2>&1; echo "parity rc $? $(tail -1 $OUT0/parity.log)"
