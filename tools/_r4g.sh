#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), {a:k[a] for a in ('post_fwd','post_mid','post_bwd','wgrad_fused','attn_bwd')})"; }
for cfg in "DR4SR_X=0" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_nont.so" "DR4SR_X=0" "DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_nont.so"; do
  echo "== $cfg"
  for b in 8192 32768; do echo -n "toys$b "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch $b --steps 60"; done
  echo -n "dense8192 "; env ${cfg//,/ } bash -c "$(declare -f run); run --batch 8192 --steps 30 --dense"
done
