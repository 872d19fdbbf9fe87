#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
( time timeout 1500 python -m pytest tests/test_gpu_meta.py tests/test_gpu_trained.py tests/test_gpu_r2_paths.py tests/test_gpu_r3_paths.py tests/test_gpu_api.py -q -x -s -k "hyper or meta or Meta" ) 2>&1 | grep -E "error|passed|failed|rel|Error" | tail -20
for i in 1 2; do timeout 300 python bench.py --model metamodel --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('metamodel', d['ms_per_step'], d['value'], d.get('outer_step_ms'))"; done
