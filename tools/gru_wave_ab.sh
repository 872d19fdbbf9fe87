#!/bin/bash
# A/B of the GRU4Rec two-layer wavefront knobs: ms per step of the default bench under each setting (settings = args, "A=1,B=2" form)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1: "; env $(echo $1 | tr ',' ' ') timeout 200 python bench.py --model gru4rec --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), d.get('final_loss'))"; }
for e in "$@"; do run "$e"; done
