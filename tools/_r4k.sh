#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
( time timeout 900 python -m pytest tests/test_gpu_fmlp.py tests/test_gpu_r2_paths.py tests/test_gpu_meta.py -q -x -k "fmlp or FMLP" ) 2>&1 | grep -E "passed|failed|Error" | tail -4
for v in "DR4SR_X=0" "DR4SR_FMLP_WGRAD_WIDE=1,DR4SR_FMLP_WGRAD_GW=48" "DR4SR_FMLP_WGRAD_GW=16" "DR4SR_FMLP_WGRAD_GW=24" "DR4SR_FMLP_WGRAD_GW=32" "DR4SR_FMLP_WGRAD_GW=48" "DR4SR_X=0"; do echo -n "$v: "; env ${v//,/ } timeout 300 python bench.py --model fmlp --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fmlp', d['ms_per_step'], d['value'])"; done
