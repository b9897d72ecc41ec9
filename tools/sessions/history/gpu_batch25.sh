#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_VARIANTS="new LEARNNOPREFETCH TPW1 TPW1NP TPW4" bash tools/ab_lib.sh "ising10m_learn ising1m_learn" 100
