#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_gpu_gru.py tests/test_gpu_r2_paths.py tests/test_gpu_r3_paths.py tests/test_gpu_api.py tests/test_gpu_meta.py tests/test_gpu_parity.py -q -x -k "gru or GRU or learns" ) 2>&1 | tail -8
for v in "DR4SR_X=0" "DR4SR_WGRAD_F32=1" "DR4SR_X=0" "DR4SR_WGRAD_F32=1"; do echo -n "$v: "; env $v timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
export ROUND=4
bash tools/trace_one.sh gru4rec --model gru4rec --steps 60
cat $O/timeline_gru4rec.txt | head -24
