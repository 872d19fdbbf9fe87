#!/bin/bash
# A/B of attention launch forms at B = 8192 toys: ms per step and the attention kernels' us per step under each setting ("A=1,B=2" form)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1: "; env $(echo $1 | tr ',' ' ') timeout 200 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --batch ${BATCH:-8192} --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), 'attn_fwd', k.get('attn_fwd'), 'attn_bwd', k.get('attn_bwd'))"; }
for e in "$@"; do run "$e"; done
