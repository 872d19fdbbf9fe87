#!/bin/bash
# A/B of env settings ("A=1,B=2" form) on the default bench (B = 256; ARGS adds bench flags, e.g. ARGS="--embed-dim 128"): ms per step and per-kernel us
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1: "; env $(echo $1 | tr ',' ' ') timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --steps 300 $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), round(d['value']), d['kernel_us_per_step'], 'loss', round(d['final_loss'],5))"; }
for e in "$@"; do run "$e"; done
