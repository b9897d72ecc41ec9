#!/bin/bash
# Round 5, GPU session 37 (runs ON THE GPU BOX): rocprofv3 --kernel-trace --stats of the small entry-parallel
# workloads at the round's last library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for WL in lr5m_learn lr5m boolw4m_learn boolw4m; do bash tools/trace_only.sh $WL 100 2>&1 | grep -E "k_gibbs_ep|k_learn_ep" | head -2; done
