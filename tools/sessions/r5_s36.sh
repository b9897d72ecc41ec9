#!/bin/bash
# Round 5, GPU session 36 (runs ON THE GPU BOX): rocprofv3 --kernel-trace --stats of the two 50M workloads at the
# round's last library (k_learn_ep_w4; scalar wave index) -- the kernel stats files beside their bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/trace_only.sh lr50m_learn 10
bash tools/trace_only.sh lr50m 10
