#!/bin/bash
# round-4 baseline: full GPU suite, default bench, B=8192 traces
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu_full.txt 2>&1
tail -5 $O/pytest_gpu_full.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
export ROUND=4
bash tools/trace_one.sh sasrec_B8192 --batch 8192 --steps 40
bash tools/trace_one.sh sasrec_B8192_dense --batch 8192 --steps 20 --dense
bash tools/trace_one.sh sasrec_B256 --steps 100
head -30 $O/kernels_sasrec_B8192.txt
