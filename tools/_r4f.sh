#!/bin/bash
# wave counts of the wave-tile kernels per batch size (rounds of tiles per wave), and the d = 128 at-scale profile
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), {a:k[a] for a in ('embqkv_fwd','post_fwd','post_mid','post_bwd','qkv_embed_bwd')})"; }
for cfg in "DR4SR_X=0" "DR4SR_WT_MID_WAVES=12" "DR4SR_WT_MID_WAVES=16" "DR4SR_WT_FWD_WAVES=16,DR4SR_WT_BWD_WAVES=16" "DR4SR_WT_FWD_WAVES=8,DR4SR_WT_BWD_WAVES=8" "DR4SR_WT_EMB_WAVES=12" "DR4SR_WT_EMB_WAVES=8"; do
  echo "== $cfg"
  for b in 4096 8192 16384 32768; do echo -n "toys$b "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch $b --steps 60"; done
  echo -n "dense8192 "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch 8192 --steps 30 --dense"
done
export ROUND=4
bash tools/trace_one.sh sasrec_d128_B8192 --batch 8192 --embed-dim 128 --steps 30
head -24 $O/kernels_sasrec_d128_B8192.txt
