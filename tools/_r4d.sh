#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
DR4SR_DP_BACKEND=gloo DP_CL_TAIL=20 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/dp_cl_check.py 2>&1 | grep -E "DP_CL|Error|error|File|assert" | head -40
DR4SR_DP_BACKEND=gloo DP_CL_TAIL=37 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 tools/dp_cl_check.py 2>&1 | grep -E "DP_CL|Error|error|File|assert" | head -40
( time timeout 900 python -m pytest tests/test_gpu_r2_paths.py -q -k "RECOMPUTE" ) 2>&1 | tail -5
