#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_cl.py tests/test_gpu_meta.py -x -q --durations=8 ) > $O/pytest_cl_meta.txt 2>&1
tail -25 $O/pytest_cl_meta.txt
( time timeout 900 python -m pytest tests/test_gpu_api.py -q -k "other_models or topk" tests/test_gpu_parity.py -k "topk or other_models" tests/test_gpu_gru.py tests/test_gpu_r3_paths.py -k "topk or other_models or gru or GRU" ) > $O/pytest_misc.txt 2>&1
tail -15 $O/pytest_misc.txt
for i in 1 2; do timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
