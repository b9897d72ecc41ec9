#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 tests/multirank_worker.py grid gibbs p2p 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo" | tail -15
