#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_cl.py tests/test_gpu_meta.py -q --durations=8 ) > $O/pytest_cl_meta.txt 2>&1
tail -25 $O/pytest_cl_meta.txt
( time timeout 1500 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_r3_paths.py tests/test_gpu_trained.py tests/test_gpu_parity.py -q -m gpu -x --durations=5 ) > $O/pytest_paths.txt 2>&1
tail -12 $O/pytest_paths.txt
bash tools/bench_b8192.sh
for i in 1 2; do timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
