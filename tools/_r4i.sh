#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 6 3 4 8 12 6; do echo -n "GW=$v: "; DR4SR_GRU_WGRAD_GW=$v timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
