#!/bin/bash
# A/B of the wave-tile kernels (csrc/linear_wave.hip) against the 256-thread tile kernels at B = 8192: step time + per-kernel launch times
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), {a: k[a] for a in ('embqkv_fwd','post_fwd','post_mid','post_bwd','qkv_embed_bwd','wgrad_fused','attn_fwd','attn_bwd','adam') if a in k})"; }
for cfg in ${WT_CFGS:-"DR4SR_NO_WAVE_TILES=1" "DR4SR_WT_WAVES=12" "DR4SR_WT_WAVES=16" "DR4SR_EXACT_F32=1" "DR4SR_EXACT_F32=1,DR4SR_WT_WAVES=16"}; do
  echo "== $cfg"
  echo -n "toys8192  "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch 8192 --steps 100"
  echo -n "dense8192 "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch 8192 --steps 40 --dense"
done
