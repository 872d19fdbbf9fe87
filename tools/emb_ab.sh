#!/bin/bash
# A/B of env settings ("A=1,B=2" form) at BATCH (default 8192; DENSE=1: all-50 rows): ms per step and the embedding-stage / tile kernels' us per step
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1: "; env $(echo $1 | tr ',' ' ') timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --batch ${BATCH:-8192} --steps ${STEPS:-100} ${DENSE:+--dense} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), {x: k.get(x) for x in ('embqkv_fwd','post_fwd','post_mid','post_bwd','qkv_embed_bwd','wgrad_fused','attn_fwd','attn_bwd','adam')}, 'loss', round(d['final_loss'],5))"; }
for e in "$@"; do run "$e"; done
