#!/bin/bash
# Runs ON THE GPU BOX: what the driver runs at round end -- the whole -m gpu suite and smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 3000 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/full_pytest.log 2>&1
echo "pytest rc $?"; tail -48 gpurun_out/full_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
