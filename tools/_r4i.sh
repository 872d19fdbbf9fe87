#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "DR4SR_X=0" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_B.so" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_C.so" "DR4SR_X=0" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_B.so" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_C.so"; do echo -n "${v##*/}: "; env $v timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
