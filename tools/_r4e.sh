#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_gpu_api.py -q -k "optimizer" ) 2>&1 | tail -12
( time timeout 1400 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu_full2.txt 2>&1
tail -22 $O/pytest_gpu_full2.txt
